#!/bin/bash
# hardware-queue mapping: GPU_MAX_HW_QUEUES x side lanes; fp32 graph and bf16 auto
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp39; mkdir -p $out
for dt in fp32 bf16; do for rep in 1 2; do for arm in q4_ss0 q8_ss0 q8_ss1 q2_ss0 q16_ss0; do
  q=${arm%%_*}; q=${q#q}; ss=${arm##*ss}
  mode=graph; [ $dt = bf16 ] && mode=auto
  GPU_MAX_HW_QUEUES=$q CTL_SIDE_STREAM=$ss timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode $mode --dtype $dt > $out/b_${dt}_${arm}_$rep.json 2> $out/b_${dt}_${arm}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_${arm}_$rep.json").read().strip().splitlines()[-1])
    print("$dt $arm $rep: %.1f slices/s %.2f ms mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}))
except Exception as e:
    print("$dt $arm $rep FAILED", e)
PY
done; done; done | tee $out/ab.txt
