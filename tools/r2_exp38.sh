#!/bin/bash
# side lanes default: does CTL_SIDE_STREAM=1 in the eager warm-up steps cost the graph replays?  fp32 graph, bf16 auto, 3 reps each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp38; mkdir -p $out
for dt in fp32 bf16; do for rep in 1 2 3; do for ss in 1 0; do
  mode=graph; [ $dt = bf16 ] && mode=auto
  CTL_SIDE_STREAM=$ss timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode $mode --dtype $dt > $out/b_${dt}_ss${ss}_$rep.json 2> $out/b_${dt}_ss${ss}_$rep.err
  python3 - <<PY
import json
d = json.loads(open("$out/b_${dt}_ss${ss}_$rep.json").read().strip().splitlines()[-1])
print("$dt ss$ss $rep: %.1f slices/s %.2f ms mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}))
PY
done; done; done | tee $out/ab.txt
