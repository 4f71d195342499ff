# resident blocks per CU of the fp32 X2 conv instantiations (variant builds of ctl_conv.hip): default = X2+EPI at 2, the rest as their tile class
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x2occ
bash tools/build_variant.sh x2a "-DCTL_LB_X2=2" ctl_conv.hip > gpurun_out/x2occ/build_a.log 2>&1; tail -1 gpurun_out/x2occ/build_a.log
bash tools/build_variant.sh x2b "-DCTL_LB_SMALL_X2EPI=3" ctl_conv.hip > gpurun_out/x2occ/build_b.log 2>&1; tail -1 gpurun_out/x2occ/build_b.log
bash tools/build_variant.sh x2c "-DCTL_LB_X2EPI=3" ctl_conv.hip > gpurun_out/x2occ/build_c.log 2>&1; tail -1 gpurun_out/x2occ/build_c.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
bash tools/ab.sh gpurun_out/x2occ -r 3 "default|" "x2_plain_at_2|--lib $V/libctl_x2a.so" "small_x2epi_at_3|--lib $V/libctl_x2b.so" "x2epi_at_3_old|--lib $V/libctl_x2c.so"
