for dbg in 6 134 198; do for pc in 4 8; do echo -n "dbg=$dbg persist=$pc: "; CTL_PERSIST=$pc CTL_DBG=$dbg python tools/bench_conv.py child fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print({k:d[k][0] for k in ('c16-16@256','c32-32@128','c64-64@64','c128-128@32','c128-128@16')})
"; done; done
