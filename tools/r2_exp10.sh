#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp10; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
timeout 900 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_engine_gpu.py -x -q > $out/pytest_bf16.log 2>&1; tail -3 $out/pytest_bf16.log
echo "== default build"; timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -v RESULT | tee $out/conv16.txt
echo "== phase timers"; CTL_HIP_LIB=$V/libctl_tm16.so timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -v RESULT | tee $out/tm16.txt

timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err
python3 - <<PY
import json
d = json.loads(open("$out/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16: %.1f slices/s  %.2f ms  mode %s calib %s  dominant %s" % (d["value"], d["ms_per_step"], d["mode"], d["mode_calibration"], d["roofline"].get("single_stream", d["roofline"])))
PY
