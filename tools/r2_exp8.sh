#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp8; mkdir -p $out
for fb in 0 1; do
  CTL_FUSE_BNBWD=$fb timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fp32_bnbwd$fb.json 2> $out/bench_fp32_bnbwd$fb.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_fp32_bnbwd$fb.json").read().strip().splitlines()[-1])
print("CTL_FUSE_BNBWD=$fb: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}))
PY
done
