#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp17; mkdir -p $out
for sl in 768 256 192 128 96 64; do
  CTL_WGRAD_SLOTS=$sl timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sl$sl.json 2> $out/bench_sl$sl.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_sl$sl.json").read().strip().splitlines()[-1])
print("CTL_WGRAD_SLOTS=$sl: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}))
PY
done
