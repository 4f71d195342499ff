cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "stacked" 2>&1 | tail -8
bash tools/ab.sh gpurun_out/r5_ab_bf16grp -r 3 "single|--dtype bf16 --masks targeted --mode segments --set nets.GROUP_WGRAD_BF16=False" "stacked|--dtype bf16 --masks targeted --mode segments"
