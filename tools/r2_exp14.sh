#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp14; mkdir -p $out
for mt4 in 0 1; do for sp in 512 768 1024; do
  echo "== MT4=$mt4 splits cap $sp"
  CTL16_WGRAD_MT4=$mt4 CTL16_WGRAD_SPLITS=$sp timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -E "wgrad" 
done; done | tee $out/wgrad.txt
