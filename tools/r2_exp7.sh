#!/bin/bash
# A/B of store cache policies (nt) on the whole step: default vs conv-nt vs elem-nt vs both; fp32 (graph) and bf16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp7; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
for lib in default cnt ent allnt; do for dt in fp32 bf16; do
  if [ $lib = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$V/libctl_$lib.so; fi
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $dt > $out/bench_${dt}_$lib.json 2> $out/bench_${dt}_$lib.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_${dt}_$lib.json").read().strip().splitlines()[-1])
print("$lib $dt: %.1f slices/s  %.2f ms  mode %s calib %s  dominant %.1f us  losses %s" % (d["value"], d["ms_per_step"], d["mode"], {k: round(v, 2) for k, v in d["mode_calibration"].items()}, d["roofline"].get("single_stream", d["roofline"])["avg_us"], [round(x, 4) for x in d["final_losses"][:3]]))
PY
done; done
