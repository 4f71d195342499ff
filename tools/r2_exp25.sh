#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 bash tools/pmc_bench.sh bf16 --dtype bf16 2>&1 | tail -16
