"""Real concurrent timeline of the plan ops of ONE training step (events on each kernel's own stream; rocprofv3
serialises dispatches and cannot show this).  Prints per 1-ms bucket how busy each stream is and the phase boundaries."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTL_PROF_TIMELINE"] = "/tmp/ctl_timeline.txt"      # read by a -DCTL_TUNING build only: CTL_TOOL_LIB=tuning python tools/timeline.py
import torch
from _variant import use_variant
_ffi = use_variant(os.environ.get("CTL_TOOL_LIB", "tuning"))
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype=os.environ.get("TIMELINE_DTYPE", "fp32"))    # TIMELINE_MASKS=targeted for config 3
if os.environ.get("TIMELINE_STACK") is not None:      # round 6: "()" = per-pass backward; a STACK_PASSES literal otherwise
    s.stack_passes = eval(os.environ["TIMELINE_STACK"])
if os.environ.get("TIMELINE_TAIL") is not None:
    s.split_wgrad_tail = bool(int(os.environ["TIMELINE_TAIL"]))
CFG = (bench.TGT_IMG, bench.TGT_SEG) if os.environ.get("TIMELINE_MASKS") == "targeted" else (bench.DROP_IMG, bench.DROP_SEG)
clean = torch.rand(16, 1, 256, 256, device="cuda"); noisy = (clean + 0.1 * torch.randn_like(clean)).clamp(0, 1)
label = torch.randint(0, 4, (16, 256, 256), device="cuda")
for _ in range(5): s.cooperative_step(clean, label, noisy, *CFG)
torch.cuda.synchronize()
_ffi.prof_start("")
s.cooperative_step(clean, label, noisy, *CFG)
torch.cuda.synchronize()
_ffi.prof_stop()
rows = [l.split() for l in open("/tmp/ctl_timeline.txt")]
ev = [(r[0], r[1], float(r[2]), float(r[3])) for r in rows]
streams = sorted(set(e[1] for e in ev), key=lambda st: min(e[2] for e in ev if e[1] == st))
end = max(e[3] for e in ev)
print(f"{len(ev)} bracketed plan ops (conv family + element-wise; losses / masks / Adam / torch ops are not bracketed), span {end:.2f} ms "
      f"(event bracketing slows the step), streams: {len(streams)}")
B = 0.5
nb = int(end / B) + 1
busy = {st: [0.0] * nb for st in streams}
for name, st, a, b in ev:
    i = int(a / B)
    while a < b:
        nxt = min(b, (i + 1) * B)
        busy[st][i] += nxt - a
        a = nxt; i += 1
print("bucket(ms)  " + "  ".join(f"stream{k}" for k in range(len(streams))))
for i in range(nb):
    print(f"{i*B:6.1f}      " + "  ".join(f"{100*busy[st][i]/B:6.0f}%" for st in streams))
tot = {st: sum(busy[st]) for st in streams}
print("plan-op busy per stream (ms):", {f"stream{k}": round(tot[st], 2) for k, st in enumerate(streams)})
# round 6: where each stream sits idle for more than 0.1 ms (between which plan ops), and the overlap of the two chains
for k, st in enumerate(streams):
    ops_ = sorted([e for e in ev if e[1] == st], key=lambda e: e[2])
    print(f"stream{k}: first op at {ops_[0][2]:.2f} ms, last ends {ops_[-1][3]:.2f} ms, {len(ops_)} ops")
    for a, b in zip(ops_, ops_[1:]):
        if b[2] - a[3] > 0.1:
            print(f"    idle {b[2] - a[3]:5.2f} ms from {a[3]:6.2f}: after {a[0][:60]} -> before {b[0][:60]}")
pts = sorted([(e[2], 1) for e in ev] + [(e[3], -1) for e in ev])
depth, last, both, one = 0, 0.0, 0.0, 0.0
for t, d in pts:
    if depth >= 2: both += t - last
    elif depth == 1: one += t - last
    depth += d; last = t
print(f"bracketed ops: {both:.2f} ms with >= 2 in flight, {one:.2f} ms with exactly one, {end - both - one:.2f} ms with none (of {end:.2f} ms)")
