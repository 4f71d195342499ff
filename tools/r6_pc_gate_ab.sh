#!/bin/bash
# round 6: same-box A/B of the producer / consumer conv gate (CTL_X3_PC_GATE of the -DCTL_TUNING build) on the training step and the inference volume
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_pc_gate; mkdir -p $out
lib=cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for r in 1 2 3; do
  for g in 0 1; do
    v=$(CTL_X3_PC_GATE=$g python3 bench.py --lib $lib --mode eager --no-sub-records --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -1 | python3 -c "import json,sys; h=json.loads(sys.stdin.read()); print('%.3f ms %.1f slices/s' % (h['ms_per_step'], h['value']))")
    echo "step fp32   gate=$g r$r  $v"
    v=$(CTL_X3_PC_GATE=$g python3 bench.py --lib $lib --workload inference --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; h=json.loads(sys.stdin.read()); print('%.3f ms %.1f slices/s' % (h['ms_per_step'], h['value']))")
    echo "inference   gate=$g r$r  $v"
  done
done
