# ceiling probes for the bf16 step: the step without the BatchNorm-backward elementwise launches (numbers are garbage, timing is what
# folding them into the neighbouring conv launches could reach at best)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_skip16
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/r3_skip16/build.log 2>&1; tail -1 gpurun_out/r3_skip16/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for mode in eager graph; do for m in 0 16 4 20 21; do
  CTL_SKIP_OPS=$m timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --mode $mode --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 $mode skip mask $m rep $rep: %.3f ms/step  launches %s' % (d['ms_per_step'], d['launches_per_step']['library']))
except Exception as e: print('skip mask $m FAILED', e)"
done; done; done | tee gpurun_out/r3_skip16/result.txt
