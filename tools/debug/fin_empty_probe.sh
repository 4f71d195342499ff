# what a BatchNorm finalize launch costs in the step: the real kernel / an empty kernel of the same grid / an empty one-wave kernel / no launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_fin_empty; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for cfg in "fp32|--mode graph" "fp32e|--mode eager" "bf16|--dtype bf16 --masks targeted --mode segments"; do for v in "real|0|0" "empty|1|0" "empty1|2|0" "skip|0|1"; do
  label=${cfg%%|*}; args=${cfg#*|}; IFS='|' read name fe sk <<< "$v"
  CTL_FIN_EMPTY=$fe CTL_SKIP_OPS=$sk timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records $args --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label forward finalize = $name rep $rep: %.3f ms/step  launches %s mode %s' % (d['ms_per_step'], json.load(open('bench_detail.json'))['launches_per_step']['library'], d.get('mode')))
except Exception as e: print('$label $name FAILED', e)"
done; done; done | tee $out/result.txt
