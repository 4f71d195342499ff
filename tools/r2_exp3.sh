#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp3; mkdir -p $out
cat > /tmp/mr.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import ops
n, c, h, w = 64, 128, 64, 64
grad = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
code = torch.rand(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
for _ in range(12): ops.latent_mask(grad, code, 0, c // 3)
torch.cuda.synchronize()
PY
for split in 64 128 256 512; do for slab in 16 32 64; do
  export CTL_MASK_SPLIT=$split CTL_MASK_SLAB_KB=$slab
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 /tmp/mr.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/raw/*/*kernel_stats.csv")[0]
d = {r["Name"].split("(")[0][-28:]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f)) if "mask_apply" in r["Name"] or "score_channel" in r["Name"]}
print("split $split slab $slab KB:", {k: round(v, 1) for k, v in d.items()}, "sum %.1f us" % sum(d.values()))
PY
  rm -rf $out/raw
done; done 2>&1 | tee $out/sweep.txt
