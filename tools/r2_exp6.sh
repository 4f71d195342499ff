#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp6; mkdir -p $out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for dt in fp32 bf16; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}.json 2> $out/bench_${dt}.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}.json").read().strip().splitlines()[-1])
print("$dt: %.1f slices/s  %.2f ms  mode %s calib %s  dominant %.1f us" % (d["value"], d["ms_per_step"], d["mode"], d["mode_calibration"], d["roofline"].get("single_stream", d["roofline"])["avg_us"]))
print("   mask:", {k: (round(v["us_per_call"],1), round(v["frac"],3), round(v.get("graph_replay_us_per_call",0),1)) for k, v in d["roofline_latent_mask"].items()})
PY
done
bash tools/prof_bench.sh r2_exp6_fp32 > $out/prof_fp32.txt 2>&1
grep -E "finalize|bwd_reduce|bwd_apply" $out/prof_fp32.txt
