# bf16 weight-gradient split cap (partials written by the kernels and re-read by the plan-end reduction: 254 MB per backward plan at 1024) inside the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/splits16
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/splits16/build.log 2>&1; tail -1 gpurun_out/splits16/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for m in 1024 768 512 256; do
  CTL16_WGRAD_SPLITS=$m timeout 300 python3 bench.py --dtype bf16 --mode eager --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 split cap $m rep $rep: %.3f ms/step' % d['ms_per_step'])
except Exception as e: print('$m FAILED', e)"
done; done | tee gpurun_out/splits16/result.txt
