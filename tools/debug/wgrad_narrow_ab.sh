# resident blocks per CU of the HBM-side weight gradients (1x1, first layers): CTL_WGRAD_NARROW_PERSIST = 1 / 2 / 4 in a -DCTL_TUNING build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wgrad_narrow
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/wgrad_narrow/build.log 2>&1; tail -1 gpurun_out/wgrad_narrow/build.log
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so
for rep in 1 2; do for dt in fp32 bf16; do for m in 1 2 4; do
  CTL_WGRAD_NARROW_PERSIST=$m timeout 300 python3 bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --lib $V 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d.get('roofline_families', {}).get('weight_gradients_1x1_and_first_layers', {})
    print('$dt narrow blocks/CU $m rep $rep: %.3f ms/step (%s)  narrow wgrad family %.2f ms at %.3f of HBM' % (d['ms_per_step'], d['mode'], f.get('ms_per_step', 0), f.get('hbm_frac', 0)))
except Exception as e: print('$dt $m FAILED', e)"
done; done; done | tee gpurun_out/wgrad_narrow/result.txt
