#!/bin/bash
# fused finalize: correctness (full gpu suite, fuse on) + A/B bench (graph + eager) for fp32 and bf16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp5; mkdir -p $out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
for fuse in 0 1; do for dt in fp32 bf16; do
  CTL_FUSE_FINALIZE=$fuse timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}_fuse$fuse.json 2> $out/bench_${dt}_fuse$fuse.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}_fuse$fuse.json").read().strip().splitlines()[-1])
print("fuse=$fuse $dt: %.1f slices/s  %.2f ms  mode %s calib %s  dominant %.1f us" % (d["value"], d["ms_per_step"], d["mode"], d["mode_calibration"], d["roofline"].get("single_stream", d["roofline"])["avg_us"]))
PY
done; done
