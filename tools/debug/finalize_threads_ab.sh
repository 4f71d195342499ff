# BatchNorm finalize kernels (224 launches per step, ~5.3 us each, one block per channel) with 64 / 128 threads per block instead of 256
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
for t in 64 128; do bash tools/build_variant.sh fin$t "-DCTL_FIN_THREADS=$t" ctl_elem.hip > gpurun_out/fin/build_$t.log 2>&1; tail -1 gpurun_out/fin/build_$t.log; done
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
bash tools/ab.sh gpurun_out/fin -r 2 "t256|" "t128|--lib $V/libctl_fin128.so" "t64|--lib $V/libctl_fin64.so" "bf16_t256|--dtype bf16 --mode eager" "bf16_t64|--dtype bf16 --mode eager --lib $V/libctl_fin64.so"
