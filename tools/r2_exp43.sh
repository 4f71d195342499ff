#!/bin/bash
# chain-overlap probe: tests + the RCCL world-1 case that serialised the eager chains (22.5 ms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp43; mkdir -p $out
timeout 900 python3 -m pytest tests/test_optin_paths_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; tail -6 $out/pytest.log
run() { name=$1; shift
  env MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 "$@" > $out/b_$name.json 2> $out/b_$name.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f slices/s %.2f ms mode %s calib %s overlap %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}, d.get("chain_overlap")))
except Exception as e:
    print("$name FAILED", e); print(open("$out/b_$name.err").read()[-600:])
PY
}
B="timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run eager_nodist $B --mode eager
run eager_dist $B --mode eager --force-dist
run auto_dist $B --force-dist
run bf16_auto_dist $B --force-dist --dtype bf16
run bf16_auto $B --dtype bf16
