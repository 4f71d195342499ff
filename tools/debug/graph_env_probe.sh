# HIP runtime knobs of the graph executor (strings of libamdhip64.so: DEBUG_HIP_FORCE_GRAPH_QUEUES, DEBUG_CLR_GRAPH_PACKET_CAPTURE,
# DEBUG_HIP_GRAPH_BATCH_SIZE) against the bf16 and fp32 graph replays
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/graph_env
run() {  # label, dtype, env...
  label=$1; dt=$2; shift 2
  env "$@" timeout 300 python3 bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --mode graph 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %-46s %.3f ms/step' % ('$dt', '$label', d['ms_per_step']))
except Exception as e: print('$dt $label FAILED', e)"
}
for dt in bf16 fp32; do
  run "default" $dt X=1
  run "FORCE_GRAPH_QUEUES=1" $dt DEBUG_HIP_FORCE_GRAPH_QUEUES=1
  run "FORCE_GRAPH_QUEUES=2" $dt DEBUG_HIP_FORCE_GRAPH_QUEUES=2
  run "FORCE_GRAPH_QUEUES=4" $dt DEBUG_HIP_FORCE_GRAPH_QUEUES=4
  run "PACKET_CAPTURE=0" $dt DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run "PACKET_CAPTURE=1" $dt DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run "BATCH_SIZE=16" $dt DEBUG_HIP_GRAPH_BATCH_SIZE=16
  run "BATCH_SIZE=256" $dt DEBUG_HIP_GRAPH_BATCH_SIZE=256
  run "default again" $dt X=1
done | tee gpurun_out/graph_env/result.txt
