#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp26; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
timeout 900 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_engine_gpu.py -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
echo "== 16-byte stores"; timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -E "c16|1x1|c32|s2 |up " | tee $out/st16.txt
echo "== 8-byte stores"; CTL_HIP_LIB=$V/libctl_st8.so timeout 300 python3 tools/bench_conv16.py 2>&1 | grep -E "c16|1x1|c32|s2 |up " | tee $out/st8.txt
for lib in default st8; do
  if [ $lib = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$V/libctl_$lib.so; fi
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > $out/bench_$lib.json 2> $out/bench_$lib.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_$lib.json").read().strip().splitlines()[-1])
print("$lib: %.1f slices/s  %.2f ms  mode %s calib %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}))
PY
done
