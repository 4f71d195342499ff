#!/bin/bash
# SQ instruction-mix counters of the X3 conv micro-benchmark (run via gpurun): tools/pmc_x3.sh  -> gpurun_out/pmc_x3/*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_x3; rm -rf $out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $out/sq_counters.txt
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -- python3 tools/bench_x3.py --one > $out/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv_igemm_kernel" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:36s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
