#!/bin/bash
# X3 weight gradient, timing ablations (variant builds wab<mask> of tools/build_variant.sh with -DCTL_X3W_ABLATE=<mask>; WRONG results by design):
#   1 no split arithmetic, 2 no global loads, 4 no MFMAs, 8 no A-operand LDS reads, 16 no staging stores, 31 all of them
export CTL_BENCH_X3=1 CTL_BENCH_N=16
for v in ${ABL:-0 1 2 4 8 16 31 63 64 159 191}; do
  echo "=== ablate $v"
  CTL_TOOL_LIB=wab$v python3 tools/bench_conv.py child wgrad 2>/dev/null | grep RESULT | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()[7:]); print('  '.join(f'{k}: {v[0]}' for k,v in r.items() if v))"
done
