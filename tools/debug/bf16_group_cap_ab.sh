cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2 3; do for cap in 100 50 25 200; do
  CTL16_WGRAD_GROUP_CAP=$cap timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --dtype bf16 --masks targeted --mode segments --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 group cap $cap rep $rep: %.3f ms/step' % d['ms_per_step'])
except Exception as e: print('cap $cap FAILED', e)"
done; done
