# fp32: BatchNorm-backward apply passes inside the consumers' staging -- kernel tests, then the same-box A/B of the plan switches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fp32_apply
timeout 1200 python3 -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "prologue or virtual or tail_backward" > gpurun_out/fp32_apply/pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/fp32_apply/pytest.log
bash tools/ab.sh gpurun_out/fp32_apply -r 2 "default|" "apply_bn2|--set nets.FUSE_BNAPPLY=True" "apply_both|--set nets.FUSE_BNAPPLY=True --set nets.FUSE_BNBWD=True" "bnbwd_only|--set nets.FUSE_BNBWD=True"
