#!/bin/bash
# order / side-lane check of the residual fp32 difference: new first, then old; new with CTL_SIDE_STREAM=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp37; mkdir -p $out
for rep in 1 2 3; do for tree in new new_ss0 old; do
  if [ $tree = old ]; then dir=$GRAFT_REPO_ROOT/.ab_old; else dir=$GRAFT_REPO_ROOT; fi
  ss=1; [ $tree = new_ss0 ] && ss=0
  (cd $dir && CTL_SIDE_STREAM=$ss timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode graph > $out/b_${tree}_$rep.json 2> $out/b_${tree}_$rep.err)
  python3 - <<PY
import json
d = json.loads(open("$out/b_${tree}_$rep.json").read().strip().splitlines()[-1])
print("fp32 $tree $rep: %.1f slices/s %.2f ms  step_ms %s" % (d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["step_ms"].items() if k != "note"}))
PY
done; done | tee $out/ab.txt
