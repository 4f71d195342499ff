#!/bin/bash
# session d: bf16 Dropout2d tests, fp32 conv occupancy-4 variant (lbm4), CTL_FUSE_BNBWD A/B, per-step kernel census of a graph replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_exp29; mkdir -p $out
V=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants
timeout 900 python3 -m pytest tests/test_dropout_gpu.py tests/test_bf16_engine_gpu.py -x -q -m gpu -s > $out/pytest_dropout.log 2>&1; tail -5 $out/pytest_dropout.log
grep "with dropout" $out/pytest_dropout.log

echo "== fp32 conv kernels: default vs -DCTL_LB_MID=4 (occupancy 4, 22 spilled VGPRs on the dominant form)"
for lib in default lbm4; do
  if [ "$lib" = default ]; then unset CTL_HIP_LIB; else export CTL_HIP_LIB=$V/libctl_$lib.so; fi
  for kind in fwd dgrad; do
    timeout 300 python3 tools/bench_conv.py child $kind 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('RESULT '):
        d=json.loads(l[7:]); print('  $lib $kind '+'  '.join('%s %s'%(k,v[0]) for k,v in d.items()))
"
  done
done | tee $out/lbm4.txt
unset CTL_HIP_LIB

echo "== CTL_FUSE_BNBWD A/B (fp32, whole step)"
for rep in 1 2; do for f in 0 1; do
  CTL_FUSE_BNBWD=$f timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_bnbwd${f}_$rep.json 2> $out/bench_bnbwd${f}_$rep.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_bnbwd${f}_$rep.json").read().strip().splitlines()[-1])
print("FUSE_BNBWD=$f rep $rep: %.1f slices/s %.2f ms mode %s launches %s" % (d["value"], d["ms_per_step"], d["mode"], d.get("launches_per_step")))
PY
done; done | tee $out/bnbwd_ab.txt

echo "== kernel census of one graph-replay step"
rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 bench.py --steps 4 --warmup 2 --mode graph --no-cpu-baseline > $out/trace_bench.log 2>&1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$out/raw/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ticks = [i for i, r in enumerate(rows) if "step_tick_kernel" in r["Kernel_Name"]]
print("dispatches", len(rows), "step ticks", len(ticks))
# the timed region's replays: the last run of equally long periods
per = [(ticks[i + 1] - ticks[i]) for i in range(len(ticks) - 1)]
print("dispatches between ticks:", per)
best = None
for i in range(len(per) - 1, 0, -1):
    if per[i] == per[i - 1]:
        best = i
        break
if best is not None:
    seg = rows[ticks[best]:ticks[best + 1]]
    cnt = collections.Counter()
    dur = collections.Counter()
    for r in seg:
        n = r["Kernel_Name"]
        n = n.split("(")[0][:90]
        cnt[n] += 1
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    busy = sum(dur.values()) / 1e3
    print("one replay: %d dispatches, span %.2f ms, summed kernel time %.2f ms" % (len(seg), span, busy))
    foreign = {n: c for n, c in cnt.items() if not any(t in n for t in ("conv_", "bwd_", "bn_", "wgrad", "sumpool", "latent_mask", "score_", "mask_apply", "adam", "ce2d", "mse", "softmax", "onehot", "step_tick", "dropout2d", "pack_", "sigmoid", "chan_sum", "argmax", "uniform", "noise", "fin_table", "grad_sum", "axpy"))}
    print("not from the library:", json.dumps(foreign, indent=1))
    json.dump({"dispatches": len(seg), "span_ms": span, "kernel_ms": busy, "count": dict(cnt), "us": {k: round(v, 1) for k, v in dur.items()}}, open("$out/graph_step_census.json", "w"), indent=1)
    qs = collections.Counter(r.get("Queue_Id", "?") for r in seg)
    print("queues:", dict(qs))
PY
rm -rf $out/raw
