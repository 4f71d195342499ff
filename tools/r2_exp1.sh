#!/bin/bash
# experiment: latent-mask rewrite (tests + roofline) and bf16 conv occupancy variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp1; mkdir -p $out
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "mask or latent or dropout" > $out/pytest_mask.log 2>&1; tail -3 $out/pytest_mask.log
timeout 300 python3 - > $out/mask_roofline.txt 2>&1 <<'PY'
import torch, json, bench
print(json.dumps(bench.latent_mask_roofline(torch.device("cuda", 0)), indent=1))
PY
cat $out/mask_roofline.txt | grep -E "frac|us_per_call|\"[ch]" 
V=cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
echo "== default (occ 3, persist 4)"; timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -v RESULT | tee $out/conv16_default.txt
for o in 4 5 6 8; do
  echo "== occ $o (persist 8)"; CTL_PERSIST=8 CTL_HIP_LIB=$PWD/$V/libctl_o$o.so timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -v RESULT | tee $out/conv16_o$o.txt
done
