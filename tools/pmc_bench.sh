#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of the bench workload; run via gpurun.
#   tools/pmc_bench.sh [TAG [bench args...]]   ->  gpurun_out/pmc_bench_TAG/traffic_by_kernel.json
tag=${1:-fp32}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_bench_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --mode eager "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --mode eager "$@" > $out/write.log 2>&1
python3 - <<PY
import csv, glob, collections, json
def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"$out/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
rows = []
for k in fe:
    n = len(fe[k]); f = sum(fe[k]) / n * 1024 * 2      # KB -> B; gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream
    w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1) * 1024
    rows.append((n, k, f, w))
rows.sort(reverse=True)
res = {k: {"launches": n, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w} for n, k, f, w in rows}
json.dump(res, open("$out/traffic_by_kernel.json", "w"), indent=1)
for n, k, f, w in rows[:14]: print(f"{k[:70]:70s} launches {n:5d}  fetch {f/1e6:8.1f} MB  write {w/1e6:8.1f} MB")
PY
rm -rf $out/fetch $out/write
