#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/aten_ops_in_step.py 2>&1 | tail -60
