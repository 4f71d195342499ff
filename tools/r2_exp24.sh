#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_optin_paths_gpu.py tests/test_cabi.py -x -q 2>&1 | tail -15
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['launches_per_step'])"
