#!/bin/bash
# round 6: is the NaN of `bench.py --gpus 2 --backend gloo --all-on-device0` (seen once in tests/test_bench_line.py) a flake of this round's code or older?
# usage: r6_dp_flake.sh TREE RUNS [extra bench args]   (TREE = . or .r5_tree)
tree=$1; runs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/$tree
bad=0
for i in $(seq 1 $runs); do
  out=$(python3 bench.py --gpus 2 --backend gloo --all-on-device0 --size 64 --batch 2 --steps 2 --warmup 1 --no-cpu-baseline --no-sub-records --detail-file /tmp/d_$i.json "$@" 2>&1)
  if echo "$out" | grep -q "nan, nan\|CHECK_FINITE"; then bad=$((bad+1)); echo "run $i: NaN"; echo "$out" | grep "CHECK_FINITE" | head -4 | cut -c1-400;
  elif echo "$out" | tail -1 | grep -q '"value"'; then echo "run $i: ok $(echo "$out" | tail -1 | python3 -c "import json,sys; h=json.loads(sys.stdin.read()); print(h.get('mode'), round(h['ms_per_step'],1))")";
  else bad=$((bad+1)); echo "run $i: FAILED"; echo "$out" | tail -5 | cut -c1-300; fi
done
echo "tree $tree: $bad bad of $runs"
