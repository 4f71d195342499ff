#!/bin/bash
# second chain on a high-priority stream (own hardware queue set) vs default; graph and eager; fp32 and bf16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp40; mkdir -p $out
for dt in fp32 bf16; do for rep in 1 2; do for pr in 0 -1; do
  CTL_CHAIN_PRIORITY=$pr timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode auto --dtype $dt > $out/b_${dt}_p${pr}_$rep.json 2> $out/b_${dt}_p${pr}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_p${pr}_$rep.json").read().strip().splitlines()[-1])
    print("$dt priority $pr rep $rep: %.1f slices/s %.2f ms mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}))
except Exception as e:
    print("$dt $pr $rep FAILED", e)
PY
done; done; done | tee $out/ab.txt
