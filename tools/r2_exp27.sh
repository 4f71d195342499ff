#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp27; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
for a in default abl1 abl2 abl4 abl3 abl7; do
  if [ $a = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$V/libctl_$a.so; fi
  echo "== $a (1 = no stores, 2 = no loads in the loop, 4 = no MFMA phase)"
  timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -E "c16-16@256 plain|c16-16@256 dgrad|1x1 16"
done | tee $out/ablate.txt
