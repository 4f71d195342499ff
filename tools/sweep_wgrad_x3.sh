#!/bin/bash
# X3 weight gradient: resident blocks per CU (CTL_X3W_PERSIST) x 16x16-pixel tiles (CTL_X3W_MT4), tuning build (run via gpurun)
export CTL_TOOL_LIB=tuning CTL_BENCH_X3=1
for n in 16 32; do for p in 1 2 3; do for m in 0 1; do
  echo "=== n=$n persist=$p mt4=$m"
  CTL_BENCH_N=$n CTL_X3W_PERSIST=$p CTL_X3W_MT4=$m python3 - <<'PY' 2>&1 | grep -v amdgpu
import subprocess, sys, json, os
out = subprocess.run([sys.executable, "tools/bench_conv.py", "child", "wgrad"], capture_output=True, text=True).stdout
line = [l for l in out.splitlines() if l.startswith("RESULT ")]
r = json.loads(line[0][7:]) if line else {}
print("  ".join(f"{k}: {v[0]}" for k, v in r.items() if v))
PY
done; done; done
