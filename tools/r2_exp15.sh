#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp15; mkdir -p $out
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for dt in fp32 bf16; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}.json 2> $out/bench_${dt}.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}.json").read().strip().splitlines()[-1])
print("$dt: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}))
PY
done
timeout 900 bash tools/pmc_bench.sh > $out/pmc.txt 2>&1; tail -14 $out/pmc.txt
cp gpurun_out/pmc_bench/traffic_by_kernel.json $out/ 2>/dev/null
