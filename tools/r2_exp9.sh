#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp9; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
CTL_HIP_LIB=$V/libctl_tm16.so timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -v RESULT | tee $out/tm16.txt
