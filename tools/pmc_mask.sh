#!/bin/bash
# HBM traffic of the latent-mask launch at the configured size (run via gpurun): tools/pmc_mask.sh -> gpurun_out/pmc_mask/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_mask; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 tools/bench_mask.py > $out/$c.log 2>&1
done
python3 - <<PY > $out/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(acc.items()):
    if "latent_mask" in k or "mask_apply" in k or "score_" in k:
        f = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [0])), 1) * 2048
        w = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [0])), 1) * 1024
        print(f"{k[:60]:60s} grid {g:>8s} n={len(cs.get('FETCH_SIZE', []))}: fetch {f/1e6:8.2f} MB  write {w/1e6:8.2f} MB  total {(f+w)/1e6:8.2f} MB")
PY
cat $out/summary.txt; rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
