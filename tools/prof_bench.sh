#!/bin/bash
# Kernel-time profile of the bench workload (run via gpurun): tools/prof_bench.sh TAG [bench args...]  ->  gpurun_out/prof_TAG/{stats.csv,bench.log}
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-sub-records --mode eager "$@" > $out/bench.log 2>&1
cp $out/raw/*/*kernel_stats.csv $out/stats.csv 2>/dev/null
rm -rf $out/raw
tail -1 $out/bench.log | cut -c1-200
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over the trace", tot / 1e6, "(10 two-chain steps + 7 single-stream replay steps + latent-mask roofline)")
for r in rows[:34]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:7.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
