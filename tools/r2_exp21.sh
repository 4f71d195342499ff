#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp21; mkdir -p $out
timeout 900 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_engine_gpu.py -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for fb in 0 1; do
  CTL_FUSE_BNBWD16=$fb timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > $out/bench_fb$fb.json 2> $out/bench_fb$fb.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/bench_fb$fb.json").read().strip().splitlines()[-1])
    print("CTL_FUSE_BNBWD16=$fb: %.1f slices/s  %.2f ms  mode %s calib %s losses %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}, [round(x, 4) for x in d["final_losses"]]))
except Exception as e:
    print("failed", e, open("$out/bench_fb$fb.err").read()[-400:])
PY
done
