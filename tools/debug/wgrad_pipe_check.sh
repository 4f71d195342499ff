#!/bin/bash
# pipelined fp32 weight gradient: kernel parity tests, then the single-layer timings (tools/bench_conv.py child wgrad)
out=gpurun_out/wgpipe; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -x -q -m gpu -k "wgrad or grad or prologue or dy2 or virtual" 2>&1 | tail -8 > $out/pytest.txt
cat $out/pytest.txt
CTL_TOOL_LIB=pipeall timeout 300 python tools/bench_conv.py child wgrad > $out/micro_pipe.txt 2>&1
if [ -f cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_nopipe.so ]; then
  CTL_TOOL_LIB=nopipe timeout 300 python tools/bench_conv.py child wgrad > $out/micro_nopipe.txt 2>&1
fi
python - <<'PY'
import json
def rd(f):
    try:
        l=[x for x in open(f).read().splitlines() if x.startswith("RESULT ")][0]; return json.loads(l[7:])
    except Exception as e: return {}
a=rd("gpurun_out/wgpipe/micro_pipe.txt"); b=rd("gpurun_out/wgpipe/micro_nopipe.txt")
for k in a:
    print(f"{k:16s} pipe {a[k][:2] if a[k] else None}   single-image {b.get(k)[:2] if b.get(k) else None}")
PY
if [ -f cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tmw.so ]; then
  CTL_TOOL_LIB=tmw timeout 200 python tools/bench_conv.py child wgrad > $out/phase_pipe.txt 2>&1
  python - <<'PY'
import json
l=[x for x in open("gpurun_out/wgpipe/phase_pipe.txt").read().splitlines() if x.startswith("RESULT ")]
if l:
    r=json.loads(l[0][7:])
    for k in ("c64-64@64","c128-128@32","c16-16@256","c64-64@32"): print("phase", k, r[k])
PY
fi
FP32_ONLY=1 CTL_TOOL_LIB=pipeall timeout 200 python tools/debug/wgrad_dy2_micro.py > $out/dy2_pipe.txt 2>&1
FP32_ONLY=1 CTL_TOOL_LIB=nopipe timeout 200 python tools/debug/wgrad_dy2_micro.py > $out/dy2_nopipe.txt 2>&1
echo "-- dy2 micro, pipelined"; grep fp32 $out/dy2_pipe.txt; echo "-- dy2 micro, single image"; grep fp32 $out/dy2_nopipe.txt
