#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp28; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
for a in default abl1 abl2 abl4 abl7; do
  if [ $a = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$V/libctl_$a.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 tools/bench_conv16.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/raw/*/*kernel_stats.csv")[0]
rows = {r["Name"]: r for r in csv.DictReader(open(f))}
def g(sub):
    for n, r in rows.items():
        if sub in n: return "%.1f us (min %.1f) x%s" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"])
    return "-"
print("$a: 3x3 FAST1 nt1 (plain+stats, dgrad):", g("conv_igemm_bf16_kernel<3, 1, 0, 4, 32, 1, 1, true>"), "| 1x1:", g("conv_igemm_bf16_kernel<1, 1, 0, 4, 32, 1, 1, true>"))
PY
  rm -rf $out/raw
done | tee $out/ablate_rocprof.txt
