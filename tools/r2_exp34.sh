#!/bin/bash
# session h: consumer-side BatchNorm finalize: kernel tests, opt-in test, A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp34; mkdir -p $out
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "consumer_side or fused_bn" > $out/pytest_k.log 2>&1; tail -12 $out/pytest_k.log
timeout 900 python3 -m pytest tests/test_optin_paths_gpu.py -x -q -m gpu > $out/pytest_optin.log 2>&1; tail -12 $out/pytest_optin.log
for dt in fp32 bf16; do for fc in 0 1 0 1; do
  CTL_FUSE_CONSUMER=$fc timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}_fc$fc.json 2> $out/bench_${dt}_fc$fc.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/bench_${dt}_fc$fc.json").read().strip().splitlines()[-1])
    print("$dt FUSE_CONSUMER=$fc: %.1f slices/s %.2f ms mode %s calib %s launches %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}, d["launches_per_step"]["library"]))
except Exception as e:
    print("$dt FUSE_CONSUMER=$fc FAILED", e); print(open("$out/bench_${dt}_fc$fc.err").read()[-1500:])
PY
done; done | tee $out/consumer_ab.txt
