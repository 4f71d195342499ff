#!/bin/bash
# Round-2 GPU session (run via gpurun): the -m gpu suite, the driver's bench command (fp32 headline), the bf16 config-3 bench,
# then rocprofv3 kernel stats of both.  Everything lands in gpurun_out/r2_$TAG/.
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_$tag; mkdir -p $out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $out/pytest_gpu.log
  tail -3 $out/pytest_gpu.log
fi
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_fp32.json 2> $out/bench_fp32.err; echo "bench fp32 exit $?"
cut -c1-400 $out/bench_fp32.json
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --dtype bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err; echo "bench bf16 exit $?"
cut -c1-400 $out/bench_bf16.json
timeout 600 bash tools/prof_bench.sh r2_${tag}_fp32 > $out/prof_fp32.txt 2>&1
timeout 600 bash tools/prof_bench.sh r2_${tag}_bf16 --dtype bf16 > $out/prof_bf16.txt 2>&1
head -30 $out/prof_bf16.txt
