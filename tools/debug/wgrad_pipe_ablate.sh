#!/bin/bash
# timing ablations of the pipelined weight-gradient loop (wrong results by construction): 1 = no staging / loads, 2 = no operand reads either,
# 3 = no ties between the MFMA pairs and their pieces either
for v in pipeall ablm1 ablm2 abl1; do
  CTL_TOOL_LIB=$v timeout 200 python tools/bench_conv.py child wgrad > gpurun_out/abl_$v.txt 2>&1
done
python - <<'PY'
import json
for v in ("pipeall","ablm1","ablm2","abl1"):
    l=[x for x in open(f"gpurun_out/abl_{v}.txt").read().splitlines() if x.startswith("RESULT ")]
    if not l: print(v, "FAILED"); continue
    r=json.loads(l[0][7:])
    print(f"{v:8s}", "  ".join(f"{k} {r[k][0]:5.1f}" for k in ("c64-64@64","c128-128@32","c16-16@256","c32-32@128","c64-64@32")))
PY
