#!/bin/bash
# Step time of {fp32 dropout, bf16 targeted} x {graph, eager} x {two launch chains, one}: what the second chain and the graph replay each buy
# (run via gpurun) -> gpurun_out/stream_modes.txt
cd $GRAFT_REPO_ROOT; out=gpurun_out/stream_modes.txt; : > $out
for dt in fp32 bf16; do for mode in graph eager; do for ss in "" "--single-stream"; do
  python3 bench.py --dtype $dt --mode $mode $ss --steps 30 --warmup 5 --no-cpu-baseline --no-sub-records 2>/dev/null | tail -1 | python3 -c "
import json,sys; h=json.loads(sys.stdin.read()); print('$dt $mode ${ss:-two-chains}: %.3f ms/step  %.1f slices/s' % (h['ms_per_step'], h['value']))" | tee -a $out
done; done; done
