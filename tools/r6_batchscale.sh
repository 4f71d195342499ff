#!/bin/bash
# round 6: what would stacking the standard and hard passes be worth at best?  The step at bs32 (every pass n = 32, the STN pairs n = 64)
# against the step at bs16, two launch chains and one: ms per slice.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_batchscale; mkdir -p $out
for dt in fp32 bf16; do
  for b in 16 32; do
    for ss in "" "--single-stream"; do
      tag="${dt}_b${b}${ss:+_single}"
      timeout 600 python3 bench.py --batch $b --dtype $dt --mode eager --no-sub-records --no-cpu-baseline --steps 20 --warmup 5 $ss --detail-file $out/detail_$tag.json 2> $out/$tag.err | tail -1 > $out/$tag.json
      python3 - $out/$tag.json $tag <<'PY'
import json, sys
h = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s %8.1f slices/s %8.3f ms/step  launches %s" % (sys.argv[2], h["value"], h["ms_per_step"], h.get("launches_per_step")))
PY
    done
  done
done
