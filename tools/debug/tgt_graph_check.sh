#!/bin/bash
# targeted masks, fp32: graph replay and live streams with the capture-aware choice of the backward form (solver._cooperative_step)
out=gpurun_out/tgt_graph2; mkdir -p $out
for m in graph eager; do for i in 1 2; do
  python bench.py --masks targeted --mode $m --steps 40 --warmup 10 --no-sub-records --no-cpu-baseline > $out/tgt_${m}_$i.json 2> $out/tgt_${m}_$i.err
done; done
python bench.py --mode graph --steps 40 --warmup 10 --no-sub-records --no-cpu-baseline > $out/drop_graph.json 2> $out/drop_graph.err
python - <<'PY' | tee gpurun_out/tgt_graph2/ab.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/tgt_graph2/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
python -m pytest tests/test_graph_gpu.py tests/test_stock_loop_gpu.py -x -q -m gpu 2>&1 | tail -3
