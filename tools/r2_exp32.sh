#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp32; mkdir -p $out
CTL_SIDE_STREAM=1 timeout 300 python3 tools/debug/side_capture_probe.py > $out/probe.log 2>&1; tail -12 $out/probe.log
if grep -q "Segmentation" $out/probe.log; then
  CTL_SIDE_STREAM=1 timeout 600 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args python3 tools/debug/side_capture_probe.py > $out/gdb.log 2>&1; grep -n "SIGSEGV" -A12 $out/gdb.log | head -30
else
  bash tools/r2_exp31.sh
fi
