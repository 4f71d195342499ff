cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh tuning "-DCTL_TUNING" > /dev/null 2>&1
for p in 1 2 3; do echo "== CTL_WGRAD_PERSIST=$p"; CTL_WGRAD_PERSIST=$p CTL_TOOL_LIB=tuning python3 tools/bench_conv.py child wgrad 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d = json.loads(l[7:])
        print('  '.join('%s %s' % (k, v[0] if v else None) for k, v in d.items()))"; done
