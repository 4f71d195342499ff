#!/bin/bash
# why is the EAGER step 4 ms slower once RCCL is initialised (world of 1)?  with / without the gradient hook, hardware-queue count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp42; mkdir -p $out
run() { # name, env..., then args
  name=$1; shift
  env MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 "$@" > $out/b_$name.json 2> $out/b_$name.err
  python3 - <<PY
import json
try:
    lines = [l for l in open("$out/b_$name.json").read().splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    last = open("$out/b_$name.json").read().strip().splitlines()[-1][:40]
    print("$name: %.1f slices/s %.2f ms mode %s calib %s | last stdout line: %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}, last))
except Exception as e:
    print("$name FAILED", e); print(open("$out/b_$name.err").read()[-600:])
PY
}
B="timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run eager_nodist $B --mode eager
run eager_dist $B --mode eager --force-dist
run eager_dist_nohook CTL_BENCH_NO_HOOK=1 $B --mode eager --force-dist
run eager_dist_q8 GPU_MAX_HW_QUEUES=8 $B --mode eager --force-dist
run eager_dist_1stream CTL_TWO_STREAMS=0 $B --mode eager --force-dist
run auto_dist $B --force-dist
