#!/bin/bash
# Gaps between consecutive kernels of the step inside hipGraph replays (rocprofv3 kernel trace; run via gpurun) -> gpurun_out/graph_gaps/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/graph_gaps; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 6 --warmup 2 --mode ${1:-graph} --no-cpu-baseline --no-sub-records > $out/bench.log 2>&1
python3 - <<PY | tee $out/summary.txt
import csv, glob, collections, statistics
rows = []
for f in glob.glob("$out/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:50]))
rows.sort()
print(len(rows), "kernels traced; queues:", collections.Counter(r[2] for r in rows).most_common(6))
byq = collections.defaultdict(list)
for r in rows: byq[r[2]].append(r)
for q, rs in byq.items():
    if len(rs) < 500: continue
    gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1)]
    small = [g for g in gaps if -50 < g < 30]
    dur = [(r[1] - r[0]) / 1e3 for r in rs]
    print(f"queue {q}: {len(rs)} kernels, duration median {statistics.median(dur):.1f} us mean {statistics.mean(dur):.1f}; gap to the next kernel on the queue: median {statistics.median(small):.2f} us, mean {statistics.mean(small):.2f} us, "
          f"p90 {sorted(small)[int(0.9 * len(small))]:.2f}; share of gaps > 1 us: {sum(g > 1 for g in small) / len(small):.2f}")
# whole-trace busy: union of intervals vs span over a window in the middle (steady state)
lo, hi = rows[len(rows) // 3][0], rows[2 * len(rows) // 3][0]
win = [r for r in rows if lo <= r[0] < hi]
ev = sorted([(r[0], 1) for r in win] + [(r[1], -1) for r in win])
busy = 0; depth = 0; last = None; two = 0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: two += t - last
    depth += d; last = t
print(f"middle third of the trace: {len(win)} kernels over {(hi - lo) / 1e6:.2f} ms; some kernel running {busy / (hi - lo):.3f} of the time, two or more {two / (hi - lo):.3f}; sum of durations / span {sum(r[1] - r[0] for r in win) / (hi - lo):.3f}")
PY
rm -rf $out/t
