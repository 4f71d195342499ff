cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_tl
bash tools/build_variant.sh tuning "-DCTL_TUNING" > gpurun_out/r3_tl/build.log 2>&1; tail -1 gpurun_out/r3_tl/build.log
CTL_TOOL_LIB=tuning timeout 300 python3 tools/timeline.py > gpurun_out/r3_tl/timeline_fp32_dropout.txt 2>&1; tail -50 gpurun_out/r3_tl/timeline_fp32_dropout.txt
cp /tmp/ctl_timeline.txt gpurun_out/r3_tl/raw_fp32_dropout.txt
