#!/bin/bash
# runtime knobs for launch latency: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory), fp32 auto and bf16 auto
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp45; mkdir -p $out
for dt in fp32 bf16; do for rep in 1 2; do for arm in default kernarg1 kernarg0; do
  case $arm in default) E="";; kernarg1) E="HIP_FORCE_DEV_KERNARG=1";; kernarg0) E="HIP_FORCE_DEV_KERNARG=0";; esac
  env $E timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/b_${dt}_${arm}_$rep.json 2> $out/b_${dt}_${arm}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_${arm}_$rep.json").read().strip().splitlines()[-1])
    print("$dt $arm $rep: %.1f slices/s %.2f ms mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}))
except Exception as e:
    print("$dt $arm $rep FAILED", e)
PY
done; done; done | tee $out/ab.txt
