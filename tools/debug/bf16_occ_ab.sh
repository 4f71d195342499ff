# bf16 conv kernels at 4 resident blocks per CU (-DCTL16_OCC=4: 128 VGPRs) against the default 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bf16_occ
bash tools/build_variant.sh occ4 "-DCTL16_OCC=4" ctl_conv_bf16.hip > gpurun_out/bf16_occ/build.log 2>&1; tail -2 gpurun_out/bf16_occ/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_occ4.so
bash tools/ab.sh gpurun_out/bf16_occ -r 2 "occ3|--dtype bf16 --mode eager" "occ4|--dtype bf16 --mode eager --lib $V"
