# resident blocks per CU of the (matrix-bound) weight gradients inside the step, re-checked after the element-wise passes left the backward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wgrad_persist
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/wgrad_persist/build.log 2>&1; tail -1 gpurun_out/wgrad_persist/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for m in 1 2; do
  CTL_WGRAD_PERSIST=$m timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d.get('roofline_families', {}).get('weight_gradients_3x3', {})
    print('fp32 wgrad blocks/CU $m rep $rep: %.3f ms/step (%s)  3x3 wgrad family %.2f ms at %.3f of the MFMA peak' % (d['ms_per_step'], d['mode'], f.get('ms_per_step', 0), f.get('mfma_frac', 0)))
except Exception as e: print('$m FAILED', e)"
done; done | tee gpurun_out/wgrad_persist/result.txt
