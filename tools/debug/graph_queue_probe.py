"""Debug: which hardware queues do the branches of the captured step's replay get?
   GPU_MAX_HW_QUEUES=8 python tools/debug/graph_queue_probe.py K [before|after]
K extra streams are touched before the first capture ("before") or one by one with a re-capture each ("after"); then the replay is timed
on 8 different LAUNCH streams.  Round-2 findings (profiles/r2_hw_queue_sharing.txt): with 8 queues the replay takes 27-31 ms on launch
streams 0-2 and 18.2 ms on streams 3-7; extra streams in front of the FIRST instantiation move the assignment, re-capturing does not (the
runtime's branch streams are created once); a two-branch probe graph of idle kernels does not predict the real graph; timing the real
replay on several launch streams inside the product segfaulted in the runtime with the default 4 queues -- not shipped (this script
dies the same way behind the 3rd-4th launch stream when GPU_MAX_HW_QUEUES is left at its default).  CTL_DTYPE=bf16 PROBE_MASKS=targeted:
config 3, whose replay takes 12.5-12.65 ms on every launch stream (the eager step: 11.8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
when = sys.argv[2] if len(sys.argv) > 2 else "before"
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)            # CTL_DTYPE=bf16 PROBE_MASKS=targeted: config 3
CFG = (bench.TGT_IMG, bench.TGT_SEG) if os.environ.get("PROBE_MASKS") == "targeted" else (bench.DROP_IMG, bench.DROP_SEG)
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, torch.device("cuda"))
for _ in range(3):
    s.cooperative_step(clean, label, noisy, *CFG)
torch.cuda.synchronize()
extra = []
def make(n):
    for _ in range(n):
        st = torch.cuda.Stream()
        _ffi.check(_ffi.lib.ctl_spin(1, st.cuda_stream), "spin")
        extra.append(st)
    torch.cuda.synchronize()
if when == "before":
    make(k)
g = CooperativeStepGraph(s, *CFG)
g(clean, label, noisy)
torch.cuda.synchronize()
def timeit(n=6):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        g(clean, label, noisy)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n
print(f"K={k} ({when}): replay {timeit():.2f} ms", flush=True)
if when == "after":          # re-capture after creating the streams
    for kk in range(1, k + 1):
        make(1)
        g.entries.clear()
        g(clean, label, noisy)
        print(f"   re-captured behind {kk} extra stream(s): replay {timeit():.2f} ms", flush=True)
# which LAUNCH stream gives the real replay its overlap?
e = next(iter(g.entries.values()))
cur = torch.cuda.current_stream()
for i in range(8):
    st = cur if i == 0 else torch.cuda.Stream()
    st.wait_stream(cur)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(st):
        for _ in range(4):
            e.graph.replay()
    torch.cuda.synchronize()
    print(f"   launch stream {i}: replay {1e3 * (time.perf_counter() - t) / 4:.2f} ms", flush=True)
