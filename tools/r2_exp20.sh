#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CTL_SIDE_STREAM=1 timeout 300 python3 -X faulthandler bench.py --steps 3 --warmup 2 --no-cpu-baseline --mode eager 2>&1 | tail -25
