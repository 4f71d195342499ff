#!/bin/bash
# session g: full GPU suite with the eager side lanes on by default + eager calibration A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp33; mkdir -p $out
timeout 1800 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
for dt in fp32 bf16; do for ss in 1 0 1 0; do
  CTL_SIDE_STREAM=$ss timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}_ss$ss.json 2> $out/bench_${dt}_ss$ss.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}_ss$ss.json").read().strip().splitlines()[-1])
print("$dt SIDE_STREAM=$ss: %.1f slices/s %.2f ms mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}))
PY
done; done | tee $out/side_stream_auto.txt
