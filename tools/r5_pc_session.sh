#!/bin/bash
# round-5 producer/consumer conv kernels: parity + per-layer timing + step A/B (run via gpurun).  usage: tools/r5_pc_session.sh TAG [nostep]
tag=${1:-pc1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
echo "== x3 + kernel tests, default build (PC where the heuristic picks it)"
timeout 900 python3 -m pytest tests/test_x3_gpu.py tests/test_kernels_gpu.py -x -q > $out/pytest_default.log 2>&1; tail -3 $out/pytest_default.log
echo "== x3 tests, PC forced onto every eligible shape (tuning build, CTL_X3_PC_MIN_STEPS=1)"
CTL_TEST_LIB=tuning CTL_X3_PC_MIN_STEPS=1 timeout 900 python3 -m pytest tests/test_x3_gpu.py -x -q > $out/pytest_forced.log 2>&1; tail -3 $out/pytest_forced.log
echo "== per-layer: PC (default)"
timeout 600 python3 tools/bench_x3.py > $out/bench_x3_pc.txt 2>&1; grep "x3" $out/bench_x3_pc.txt | sed 's/fp32.*x3/x3/' 
echo "== per-layer: single-role kernels (tuning build, CTL_X3_PC=0)"
CTL_TOOL_LIB=tuning CTL_X3_PC=0 timeout 600 python3 tools/bench_x3.py > $out/bench_x3_old.txt 2>&1; grep "x3" $out/bench_x3_old.txt | sed 's/fp32.*x3/x3/'
if [ "$2" != "nostep" ]; then
  echo "== step A/B"
  export CTL_X3_PC=0
  bash tools/ab.sh $out/ab -r 2 "pc|" "old|--lib cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so" 2>&1 | tail -8
fi
