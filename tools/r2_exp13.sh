#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp13; mkdir -p $out
timeout 900 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_engine_gpu.py tests/test_kernels_gpu.py -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -E "bwd_" | tee $out/elem.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err
python3 - <<PY
import json
d = json.loads(open("$out/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], d["mode_calibration"]))
PY
bash tools/prof_bench.sh r2_exp13_bf16 --dtype bf16 > $out/prof_bf16.txt 2>&1; head -40 $out/prof_bf16.txt
