#!/bin/bash
# session f: side lanes under capture: test + graph-mode A/B (fp32, bf16)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp31; mkdir -p $out
timeout 900 python3 -m pytest tests/test_optin_paths_gpu.py -x -q -m gpu > $out/pytest_optin.log 2>&1; tail -15 $out/pytest_optin.log
for dt in bf16 fp32; do for ss in 0 1 0 1; do
  CTL_SIDE_STREAM=$ss timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode graph --dtype $dt > $out/bench_${dt}_ss$ss.json 2> $out/bench_${dt}_ss$ss.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/bench_${dt}_ss$ss.json").read().strip().splitlines()[-1])
    print("$dt graph SIDE_STREAM=$ss: %.1f slices/s %.2f ms  losses %s" % (d["value"], d["ms_per_step"], d.get("final_losses", d.get("losses"))))
except Exception as e:
    print("$dt graph SIDE_STREAM=$ss FAILED", e); print(open("$out/bench_${dt}_ss$ss.err").read()[-1500:])
PY
done; done | tee $out/side_stream_graph.txt
