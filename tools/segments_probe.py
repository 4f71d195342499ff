#!/usr/bin/env python3
"""Segment replay of the whole step: does the priority of the second chain's stream (or of the stream the replay is launched on) matter?
Usage: segments_probe.py [fp32|bf16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
from cooperative_training_and_latent_space_data_augmentation_amd.hipgraph import SegmentReplay

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(0)
solver = AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard", image_ch=1, num_classes=4, learning_rate=1e-4, use_gpu=True, compute_dtype=dtype)
IMG, SEG, _ = bench.MASKS["targeted" if dtype == "bf16" else "dropout"]
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, device)
for _ in range(3):
    solver.cooperative_step(clean, label, noisy, IMG, SEG)
g = CooperativeStepGraph(solver, IMG, SEG)
g(clean, label, noisy)
torch.cuda.synchronize()
e = next(iter(g.entries.values()))


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


print(f"{dtype} runtime replay: {timed(lambda: g(clean, label, noisy)):.3f} ms", flush=True)
g.replay_mode = "segments"
for emit in ("close", "open"):
    for swap in (False, True):
        e.segments = SegmentReplay(e.graph, emit=emit, swap_chains=swap)
        ts = [timed(lambda: g(clean, label, noisy)) for _ in range(3)]
        print(f"{dtype} segments, launch order by segment {emit}, chains swapped {swap}: " + " / ".join(f"{t:.3f}" for t in ts) + " ms", flush=True)
g.replay_mode = "runtime"
print(f"{dtype} runtime replay again: {timed(lambda: g(clean, label, noisy)):.3f} ms", flush=True)
