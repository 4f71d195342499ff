#!/bin/bash
# round 6: timelines (tuning build) and rocprofv3 kernel sums of the per-pass and the stacked step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_timeline; mkdir -p $out
DEF="(('image_decoder',1),('segmentation_decoder',0),('image_encoder',0))"
for dt in fp32 bf16; do
  TIMELINE_DTYPE=$dt TIMELINE_STACK="()" CTL_TOOL_LIB=tuning timeout 300 python3 tools/timeline.py > $out/timeline_${dt}_none.txt 2>&1
  TIMELINE_DTYPE=$dt TIMELINE_STACK="$DEF" CTL_TOOL_LIB=tuning timeout 300 python3 tools/timeline.py > $out/timeline_${dt}_default.txt 2>&1
  TIMELINE_DTYPE=$dt TIMELINE_STACK="$DEF" TIMELINE_TAIL=0 CTL_TOOL_LIB=tuning timeout 300 python3 tools/timeline.py > $out/timeline_${dt}_default_notail.txt 2>&1
done
tail -25 $out/timeline_fp32_none.txt; tail -25 $out/timeline_fp32_default.txt
bash tools/prof_bench.sh r6_none --set "solver.STACK_PASSES=()" > $out/prof_none.txt 2>&1
bash tools/prof_bench.sh r6_default > $out/prof_default.txt 2>&1
cp gpurun_out/prof_r6_none/stats.csv $out/stats_none.csv; cp gpurun_out/prof_r6_default/stats.csv $out/stats_default.csv
head -3 $out/prof_none.txt; head -3 $out/prof_default.txt
