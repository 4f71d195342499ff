#!/bin/bash
# more partial rows (= blocks) for the BatchNorm-backward reductions: CTL_RED_BLOCKS 512 (default) / 1024 / 2048
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp48; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
CTL_RED_BLOCKS=1024 CTL_HIP_LIB=$V/libctl_red1024.so timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "batchnorm or bwd_reduce or losses or ce2d or mse" 2>&1 | tail -2
for rep in 1 2; do for r in 512 1024 2048; do for dt in fp32 bf16; do
  if [ $r = 512 ]; then unset CTL_HIP_LIB; unset CTL_RED_BLOCKS; else export CTL_HIP_LIB=$V/libctl_red$r.so; export CTL_RED_BLOCKS=$r; fi
  mode=graph; [ $dt = bf16 ] && mode=eager
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt --mode $mode > $out/b_${dt}_${r}_$rep.json 2> $out/b_${dt}_${r}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_${r}_$rep.json").read().strip().splitlines()[-1])
    print("$dt rows<=$r rep $rep: %.1f slices/s %.2f ms" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$dt $r $rep FAILED", e); print(open("$out/b_${dt}_${r}_$rep.err").read()[-400:])
PY
done; done; done | tee $out/ab.txt
