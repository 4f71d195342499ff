# round 5: bias + prologue coefficients requested behind the first tile's loads (one wait) vs the old prologue (variant oldpro built from the previous commit)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_x3_gpu.py tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -3
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_oldpro.so
bash tools/ab.sh gpurun_out/r5_ab_prologue -r 3 "old|--mode graph --lib $V" "new|--mode graph" "old_bf16|--dtype bf16 --masks targeted --mode segments --lib $V" "new_bf16|--dtype bf16 --masks targeted --mode segments"
