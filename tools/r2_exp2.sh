#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp2; mkdir -p $out
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "mask or latent or dropout" > $out/pytest_mask.log 2>&1; tail -3 $out/pytest_mask.log
cat > /tmp/mr.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch, json, bench
print(json.dumps(bench.latent_mask_roofline(torch.device("cuda", 0)), indent=1))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 /tmp/mr.py > $out/mask_roofline.txt 2>&1
grep -E "frac|us_per_call|\"[ch]" $out/mask_roofline.txt
cp $out/raw/*/*kernel_stats.csv $out/mask_stats.csv; rm -rf $out/raw
cut -d, -f1-4,6,7 $out/mask_stats.csv | cut -c1-160 | head -12
