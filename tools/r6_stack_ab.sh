#!/bin/bash
# round 6: same-box A/B of the stacked backward (solver.STACK_PASSES variants), eager, fp32 + bf16.  usage: r6_stack_ab.sh TAG [rounds]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r6_stack_ab}; rounds=${2:-2}
out=gpurun_out/$tag; mkdir -p $out
declare -A V
V[none]="()"
V[decs]="(('image_decoder',1),('segmentation_decoder',0))"
V[default]="(('image_decoder',1),('segmentation_decoder',0),('image_encoder',0))"
V[enc_side]="(('image_decoder',1),('segmentation_decoder',0),('image_encoder',1))"
V[all_main]="(('image_decoder',0),('segmentation_decoder',0),('image_encoder',0))"
V[enc_only]="(('image_encoder',0),)"
for r in $(seq 1 $rounds); do
  for dt in fp32 bf16; do
    for v in none default default_notail enc_side all_main; do
      extra=""; vv=$v
      if [ "$v" = "default_notail" ]; then extra="--set solver.SPLIT_WGRAD_TAIL=False"; vv=default; fi
      timeout 600 python3 bench.py --dtype $dt --mode eager --no-sub-records --no-cpu-baseline --steps 30 --warmup 8 --set "solver.STACK_PASSES=${V[$vv]}" $extra --detail-file $out/detail_${dt}_${v}_$r.json 2> $out/${dt}_${v}_$r.err | tail -1 > $out/${dt}_${v}_$r.json
      python3 - $out/${dt}_${v}_$r.json "$dt $v r$r" <<'PY'
import json, sys
try:
    h = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.1f slices/s %8.3f ms/step" % (sys.argv[2], h["value"], h["ms_per_step"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    done
  done
done
