# bf16 two-tensor conv instantiations at 2 resident blocks per CU (-DCTL16_OCC_X2=2: no scratch) against 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bf16_x2occ
bash tools/build_variant.sh b16x2 "-DCTL16_OCC_X2=2" ctl_conv_bf16.hip > gpurun_out/bf16_x2occ/build.log 2>&1; tail -1 gpurun_out/bf16_x2occ/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_b16x2.so
bash tools/ab.sh gpurun_out/bf16_x2occ -r 3 "occ3|--dtype bf16 --mode eager" "occ2|--dtype bf16 --mode eager --lib $V"
