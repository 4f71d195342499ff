# ceiling probe (round 5): the step without the BatchNorm finalize launches (bit 0 forward, bit 1 backward; numbers are garbage, the timing is
# what folding them into the producing launches could reach at best).  Needs variants/libctl_tuning.so (tools/build_variant.sh tuning -DCTL_TUNING).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_skip; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for cfg in "fp32|--mode graph" "bf16|--dtype bf16 --masks targeted --mode segments"; do for m in 0 1 2 3; do
  label=${cfg%%|*}; args=${cfg#*|}
  CTL_SKIP_OPS=$m timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records $args --lib $V 2>$out/err_${label}_$m.txt | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label skip mask $m rep $rep: %.3f ms/step  launches %s mode %s' % (d['ms_per_step'], json.load(open('bench_detail.json'))['launches_per_step']['library'], d.get('mode')))
except Exception as e: print('$label skip mask $m FAILED', e)"
done; done; done | tee $out/result.txt
