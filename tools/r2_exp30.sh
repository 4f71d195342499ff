#!/bin/bash
# session e: bf16 dropout tests again, lbm4 A/B, side stream for the weight gradients (eager, fp32 + bf16)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp30; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
timeout 900 python3 -m pytest tests/test_dropout_gpu.py -x -q -m gpu -s > $out/pytest_dropout.log 2>&1; tail -3 $out/pytest_dropout.log
grep "with dropout" $out/pytest_dropout.log | grep -v print
echo "== fp32 conv kernels: default vs -DCTL_LB_MID=4"
for lib in default lbm4; do
  if [ "$lib" = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$V/libctl_$lib.so; fi
  for kind in fwd dgrad; do
    timeout 300 python3 tools/bench_conv.py child $kind 2>$out/bc_$lib.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print('  $lib $kind '+'  '.join('%s %s'%(k,v[0]) for k,v in d.items()))
"
  done
done | tee $out/lbm4.txt
unset CTL_HIP_LIB
echo "== side stream (eager mode)"
for dt in bf16 fp32; do for ss in 0 1; do
  CTL_SIDE_STREAM=$ss timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode eager --dtype $dt > $out/bench_${dt}_ss$ss.json 2> $out/bench_${dt}_ss$ss.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}_ss$ss.json").read().strip().splitlines()[-1])
print("$dt eager SIDE_STREAM=$ss: %.1f slices/s %.2f ms" % (d["value"], d["ms_per_step"]))
PY
done; done | tee $out/side_stream.txt
