#!/bin/bash
# host issue order of the two generation passes: segmentation part (main chain) before the image part (side chain)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp49; mkdir -p $out
CTL_GEN_ORDER=1 timeout 900 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "two_stream" 2>&1 | tail -3
for rep in 1 2 3; do for o in 0 1; do
  CTL_GEN_ORDER=$o timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 --mode eager > $out/b_bf16_o${o}_$rep.json 2> $out/b_bf16_o${o}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_bf16_o${o}_$rep.json").read().strip().splitlines()[-1])
    print("bf16 eager gen order $o rep $rep: %.1f slices/s %.2f ms" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$o $rep FAILED", e)
PY
done; done | tee $out/ab.txt
for o in 0 1; do CTL_GEN_ORDER=$o timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --masks targeted --mode eager | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 targeted gen order $o', round(d['value'],1), round(d['ms_per_step'],2))"; done | tee -a $out/ab.txt
