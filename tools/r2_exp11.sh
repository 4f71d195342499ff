#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp11; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
CTL_HIP_LIB=$V/libctl_tm32.so timeout 300 python3 tools/bench_conv.py child fwd 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:])
        for k,v in d.items(): print('  %-20s %s'%(k,v))
" | tee $out/tm32_fwd.txt
