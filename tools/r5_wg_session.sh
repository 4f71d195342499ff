#!/bin/bash
# round-5 producer/consumer X3 kernels: parity (natural + forced) and per-layer timing (run via gpurun).  usage: tools/r5_wg_session.sh TAG [step]
tag=${1:-wg1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
echo "== x3 tests, default build"
timeout 1200 python3 -m pytest tests/test_x3_gpu.py -x -q > $out/pytest_default.log 2>&1; tail -3 $out/pytest_default.log
echo "== x3 tests, PC forced onto every eligible shape (tuning build)"
CTL_TEST_LIB=tuning CTL_X3_PC_MIN_STEPS=1 CTL_X3_PC_MIN_G=1 CTL_X3W_PC_MIN_TILES=1 timeout 1200 python3 -m pytest tests/test_x3_gpu.py -x -q > $out/pytest_forced.log 2>&1; tail -3 $out/pytest_forced.log
echo "== wgrad per layer: PC (default)"
timeout 600 python3 tools/bench_wgrad_x3.py 2>&1 | grep -v amdgpu | tee $out/bench_wgrad_pc.txt
echo "== wgrad per layer: single-role (CTL_X3W_PC=0)"
CTL_TOOL_LIB=tuning CTL_X3W_PC=0 timeout 600 python3 tools/bench_wgrad_x3.py 2>&1 | grep -v amdgpu | tee $out/bench_wgrad_old.txt
if [ "$2" = "step" ]; then
  echo "== step A/B"
  export CTL_X3W_PC=0 CTL_X3_PC=0
  bash tools/ab.sh $out/ab -r 2 "pc|" "old|--lib cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_tuning.so" 2>&1 | tail -8
fi
