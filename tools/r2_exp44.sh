#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp44; mkdir -p $out
timeout 900 python3 -m pytest tests/test_optin_paths_gpu.py -x -q -m gpu -k captured 2>&1 | tail -30
GPU_MAX_HW_QUEUES=4 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/q4.json 2> $out/q4.err; tail -5 $out/q4.err; tail -c 600 $out/q4.json
