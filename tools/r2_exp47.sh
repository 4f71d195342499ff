#!/bin/bash
# two-chain issue order: standard D_seg -> STN forward at the head of the main chain (CTL_CHAIN_ORDER=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp47; mkdir -p $out
CTL_CHAIN_ORDER=1 timeout 900 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "two_stream or cooperative_step_vs" 2>&1 | tail -3
for rep in 1 2; do for dt in bf16 fp32; do for o in 0 1; do
  CTL_CHAIN_ORDER=$o timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt --mode eager > $out/b_${dt}_o${o}_$rep.json 2> $out/b_${dt}_o${o}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_o${o}_$rep.json").read().strip().splitlines()[-1])
    print("$dt eager order $o rep $rep: %.1f slices/s %.2f ms" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$dt $o $rep FAILED", e)
PY
done; done; done | tee $out/ab.txt
