#!/bin/bash
# bf16: weight-gradient split count end to end (kernel + plan-end reduction of the partial tensors)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp46; mkdir -p $out
for rep in 1 2; do for sp in 1024 768 512 256; do
  CTL16_WGRAD_SPLITS=$sp timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 --mode eager > $out/b_${sp}_$rep.json 2> $out/b_${sp}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${sp}_$rep.json").read().strip().splitlines()[-1])
    print("bf16 eager splits $sp rep $rep: %.1f slices/s %.2f ms" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$sp $rep FAILED", e)
PY
done; done | tee $out/ab.txt
