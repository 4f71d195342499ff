# ceiling probes: the step without certain launch families (numbers are garbage, timing is what a perfect fusion could reach)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_skip
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/r3_skip/build.log 2>&1; tail -1 gpurun_out/r3_skip/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for m in 0 1 2 3 4 8 16 31; do
  CTL_SKIP_OPS=$m timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --mode graph --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip mask $m rep $rep: %.3f ms/step  launches %s' % (d['ms_per_step'], d['launches_per_step']['library']))
except Exception as e: print('skip mask $m FAILED', e)"
done; done | tee gpurun_out/r3_skip/result.txt
