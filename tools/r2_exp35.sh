#!/bin/bash
# same-box A/B: the session-c commit (worktree .ab_old) against HEAD, graph and eager, fp32 and bf16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp35; mkdir -p $out
for dt in fp32 bf16; do for mode in graph auto; do for rep in 1 2; do for tree in old new; do
  if [ $tree = old ]; then dir=$GRAFT_REPO_ROOT/.ab_old; else dir=$GRAFT_REPO_ROOT; fi
  (cd $dir && timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode $mode --dtype $dt > $out/b_${dt}_${mode}_${tree}_$rep.json 2> $out/b_${dt}_${mode}_${tree}_$rep.err)
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_${mode}_${tree}_$rep.json").read().strip().splitlines()[-1])
    r = d["roofline"].get("single_stream") or d["roofline"]
    print("$dt $mode $tree $rep: %.1f slices/s %.2f ms mode %s calib %s dominant %.1f us" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}, r["avg_us"]))
except Exception as e:
    print("$dt $mode $tree $rep FAILED", e)
PY
done; done; done; done | tee $out/ab.txt
