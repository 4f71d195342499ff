#!/bin/bash
# bf16 auto: allocations inside the timed region after the switch back to eager; RCCL initialised in a world of 1 (queue sharing with its streams)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp41; mkdir -p $out
for arm in bf16 bf16 fp32_dist bf16_dist fp32_dist; do
  dt=${arm%%_*}; extra=""; [ "${arm##*_}" = dist ] && extra="--force-dist"
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt $extra > $out/b_$arm.json 2> $out/b_$arm.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_$arm.json").read().strip().splitlines()[-1])
    print("$arm: %.1f slices/s %.2f ms mode %s calib %s allocs %s step_ms %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}, d["device_allocs_in_timed_region"], {k: round(v, 2) for k, v in d["step_ms"].items() if k != "note"}))
except Exception as e:
    print("$arm FAILED", e); print(open("$out/b_$arm.err").read()[-800:])
PY
done | tee $out/ab.txt
