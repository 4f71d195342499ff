# bf16 step, eager and graph, twice each (TAG = output directory; extra bench flags after it)
tag=${1:-bf16b}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do for mode in eager graph; do
  timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --mode $mode "$@" 2> $out/bench_$mode.err | tail -1 > $out/bench_$mode.json
  python3 -c "
import json; d = json.loads(open('$out/bench_$mode.json').read()); print('bf16 $mode rep $rep: %.3f ms/step %.1f slices/s launches %s' % (d['ms_per_step'], d['value'], d['launches_per_step']['library']))" || tail -5 $out/bench_$mode.err
done; done
