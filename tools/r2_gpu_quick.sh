#!/bin/bash
# quick GPU check: full -m gpu suite + fp32 and bf16 bench lines (no CPU baseline)
tag=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_$tag; mkdir -p $out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for dt in fp32 bf16; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}.json 2> $out/bench_${dt}.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}.json").read().strip().splitlines()[-1])
r = d["roofline"].get("single_stream", d["roofline"])
print("$dt: %.1f slices/s  %.2f ms  mode %s calib %s  dominant %.1f us frac %.3f" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}, r["avg_us"], r["frac"]))
print("   mask:", {k: (round(v["us_per_call"],1), round(v["frac"],3), round(v.get("graph_replay_us_per_call",0),1)) for k, v in d["roofline_latent_mask"].items()})
PY
done
