#!/bin/bash
# Reproduce the driver's measurement condition: rocm-smi polled beside the bench (BENCH_r01.json pulled smi.*.json files).
# usage: tools/bench_with_smi.sh <out-prefix> [bench args...]
out=$1; shift
( while true; do rocm-smi --showuse --showmemuse --showpower --showclocks --json > /dev/null 2>&1; sleep ${SMI_PERIOD:-1}; done ) &
smi=$!
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" > ${out}.json 2> ${out}.err
kill $smi
python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), d.get('step_ms'))" ${out}.json
