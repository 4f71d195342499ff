#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp18; mkdir -p $out
run() { # name env...
  name=$1; shift
  env "$@" timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > $out/bench_$name.json 2> $out/bench_$name.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1])
print("$name: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}))
PY
}
EXTRA=""
run fp32_default X=1
run fp32_persist3 CTL_PERSIST=3 CTL_WGRAD_PERSIST=1
run fp32_persist2 CTL_PERSIST=2 CTL_WGRAD_PERSIST=1
EXTRA="--dtype bf16"
run bf16_default X=1
run bf16_wsplits512 CTL16_WGRAD_SPLITS=512
run bf16_wsplits256 CTL16_WGRAD_SPLITS=256
run bf16_persist2 CTL_PERSIST=2
