for pc in 2 3 4 5 6; do echo -n "persist=$pc : "; CTL_PERSIST=$pc python tools/bench_conv.py child fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print({k:d[k][0] for k in d})
"; done
for pc in 2 3 4; do echo -n "bench persist=$pc : "; CTL_PERSIST=$pc python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
