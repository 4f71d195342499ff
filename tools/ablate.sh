for dbg in 0 1 2 4 3 5 6 7; do for pc in 4; do echo -n "dbg=$dbg persist=$pc : "; CTL_DBG=$dbg CTL_PERSIST=$pc python tools/bench_conv.py child fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print({k:d[k][0] for k in ('c16-16@256','c64-64@64','c128-128@32','1x1 16-16@256')})
"; done; done
for pc in 2 3 6 8; do echo -n "dbg=0 persist=$pc : "; CTL_PERSIST=$pc python tools/bench_conv.py child fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print({k:d[k][0] for k in ('c16-16@256','c64-64@64','c128-128@32','1x1 16-16@256')})
"; done
