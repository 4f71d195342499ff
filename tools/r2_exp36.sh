#!/bin/bash
# same-box A/B after compiling the consumer-side path out of the default kernels; the variant build still passes its tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r2_exp36; mkdir -p $out
CTL_HIP_LIB=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_consumer.so timeout 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_optin_paths_gpu.py -x -q -m gpu -k "consumer or optin" > $out/pytest_consumer_variant.log 2>&1; tail -3 $out/pytest_consumer_variant.log
for dt in fp32 bf16; do for rep in 1 2; do for tree in old new; do
  if [ $tree = old ]; then dir=$GRAFT_REPO_ROOT/.ab_old; else dir=$GRAFT_REPO_ROOT; fi
  mode=graph; [ $dt = bf16 ] && mode=auto
  (cd $dir && timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode $mode --dtype $dt > $out/b_${dt}_${tree}_$rep.json 2> $out/b_${dt}_${tree}_$rep.err)
  python3 - <<PY
import json
try:
    d = json.loads(open("$out/b_${dt}_${tree}_$rep.json").read().strip().splitlines()[-1])
    r = d["roofline"].get("single_stream") or d["roofline"]
    print("$dt $tree $rep: %.1f slices/s %.2f ms mode %s calib %s dominant %.1f us" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d.get("mode_calibration", {}).items()}, r["avg_us"]))
except Exception as e:
    print("$dt $tree $rep FAILED", e)
PY
done; done; done | tee $out/ab.txt
