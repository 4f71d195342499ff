cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_layers
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/r3_layers/build.log 2>&1; tail -1 gpurun_out/r3_layers/build.log
CTL_TOOL_LIB=tuning CTL_PROF_SHAPES=1 timeout 300 python3 tools/debug/layer_table.py > gpurun_out/r3_layers/fp32.txt 2>&1
CTL_TOOL_LIB=tuning CTL_PROF_SHAPES=1 timeout 300 python3 tools/debug/layer_table.py bf16 > gpurun_out/r3_layers/bf16.txt 2>&1
head -75 gpurun_out/r3_layers/fp32.txt
