#!/bin/bash
# One GPU session (run via gpurun): tools/gpu_session.sh TAG [tests|notests] [pmc]
#   -m gpu suite, the driver's bench command (headline + sub-records), rocprofv3 kernel stats of the fp32 and bf16 steps, optionally the
#   PMC traffic passes.  Everything lands in gpurun_out/$TAG/ ; copy what is to be judged into profiles/.
tag=${1:-s}; tests=${2:-tests}; pmc=${3:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
if [ "$tests" = "tests" ]; then
  timeout 3000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $out/pytest_gpu.log
  tail -5 $out/pytest_gpu.log
fi
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
cp bench_detail.json $out/bench_detail.json; cp bench_detail_config3_bf16.json bench_detail_config5_inference.json bench_detail_config4_random_masks_n1.json $out/ 2>/dev/null
python3 - $out <<'PY'
import json, sys
out = sys.argv[1]
line = open(out + "/bench.json").read().strip().splitlines()[-1]
h = json.loads(line)
print("headline line: %d bytes; %.1f slices/s %.3f ms mode %s; roofline %s frac %.3f; cpu %s" % (len(line), h["value"], h["ms_per_step"], h["mode"], h["roofline"]["kernel"], h["roofline"]["frac"], h.get("cpu_baseline", {}).get("value")))
for k in ("config3_bf16", "config5_inference", "config4_random_masks_n1"): print("  ", k, h.get(k))
d = json.load(open(out + "/bench_detail.json"))
print("launches", d["launches_per_step"]["library"], "calibration", d["mode_calibration"])
for k, v in d.get("roofline_families", {}).items(): print("   family %-38s %6.2f ms/step %6.1f launches  frac %.3f (%s)" % (k, v["ms_per_step"], v["launches_per_step"], v["frac"], v["bound"]))
PY
timeout 600 bash tools/prof_bench.sh ${tag}_fp32 > $out/prof_fp32.txt 2>&1; cp gpurun_out/prof_${tag}_fp32/stats.csv $out/kernel_stats_fp32.csv
timeout 600 bash tools/prof_bench.sh ${tag}_bf16 --dtype bf16 > $out/prof_bf16.txt 2>&1; cp gpurun_out/prof_${tag}_bf16/stats.csv $out/kernel_stats_bf16.csv
head -24 $out/prof_fp32.txt
if [ "$pmc" = "pmc" ]; then
  timeout 900 bash tools/pmc_bench.sh ${tag}_fp32 > $out/pmc_fp32.txt 2>&1; cp gpurun_out/pmc_bench_${tag}_fp32/traffic_by_kernel.json $out/pmc_traffic_by_kernel_fp32.json
  tail -16 $out/pmc_fp32.txt
  timeout 900 bash tools/pmc_bench.sh ${tag}_bf16 --dtype bf16 > $out/pmc_bf16.txt 2>&1; cp gpurun_out/pmc_bench_${tag}_bf16/traffic_by_kernel.json $out/pmc_traffic_by_kernel_bf16.json
  tail -8 $out/pmc_bf16.txt
fi
