# quick bf16 iteration loop: the bf16 kernel / backward tests, then the bf16 step eager and graph (TAG = output directory)
tag=${1:-bf16q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_backward_gpu.py tests/test_bf16_engine_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $out/pytest.log
for mode in eager graph; do
  timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --mode $mode 2> $out/bench_$mode.err | tail -1 > $out/bench_$mode.json
  python3 -c "
import json; d = json.loads(open('$out/bench_$mode.json').read()); print('bf16 $mode: %.3f ms/step %.1f slices/s launches %s' % (d['ms_per_step'], d['value'], d['launches_per_step']['library']))" || tail -5 $out/bench_$mode.err
done
