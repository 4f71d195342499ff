#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp19; mkdir -p $out
run() { name=$1; shift
  env "$@" timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > $out/bench_$name.json 2> $out/bench_$name.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d["mode_calibration"].items()}))
except Exception as e:
    print("$name failed", e, open("$out/bench_$name.err").read()[-300:])
PY
}
EXTRA="--dtype bf16"
run bf16_default X=1
run bf16_side CTL_SIDE_STREAM=1
EXTRA=""
run fp32_side CTL_SIDE_STREAM=1
