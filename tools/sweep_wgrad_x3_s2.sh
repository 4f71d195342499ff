export CTL_BENCH_X3=1 CTL_BENCH_N=16 CTL_TOOL_LIB=wtune
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
  echo "=== s2 mt=$1 ntw=$2 (0 = default)"
  CTL_X3W_S2_MT=$1 CTL_X3W_S2_NTW=$2 python3 tools/bench_conv.py child wgrad 2>/dev/null | grep RESULT | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()[7:]); print('  '.join(f'{k}: {v[0]}' for k,v in r.items() if v and k.startswith('s2')))"
done
