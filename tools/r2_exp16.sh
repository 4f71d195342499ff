#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp16; mkdir -p $out
timeout 900 python3 tools/bench_configs.py --cpu 2> $out/err.txt | tee $out/other_configs.jsonl
tail -3 $out/err.txt
