#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_dropout_gpu.py -x -q 2>&1 | tail -30
