for st in 0 1 2 3 4 6; do echo -n "stagger=$st : "; CTL_DBG=$((st*256)) python tools/bench_conv.py child fwd 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print({k:d[k][0] for k in ('c16-16@256','c32-32@128','c64-64@64','c128-128@32','1x1 16-16@256')})
"; done
