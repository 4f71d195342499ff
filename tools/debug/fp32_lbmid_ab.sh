# fp32 conv kernels of the 4-fragment class at 2 resident blocks per CU instead of 3 (-DCTL_LB_MID=2), and the small class at 3 instead of 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lbmid
bash tools/build_variant.sh lbm2 "-DCTL_LB_MID=2" ctl_conv.hip > gpurun_out/lbmid/build_a.log 2>&1; tail -1 gpurun_out/lbmid/build_a.log
bash tools/build_variant.sh lbs3 "-DCTL_LB_SMALL=3" ctl_conv.hip > gpurun_out/lbmid/build_b.log 2>&1; tail -1 gpurun_out/lbmid/build_b.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
bash tools/ab.sh gpurun_out/lbmid -r 2 "default|" "mid_at_2|--lib $V/libctl_lbm2.so" "small_at_3|--lib $V/libctl_lbs3.so"
