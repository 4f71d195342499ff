#!/usr/bin/env python3
"""Does the replay time of the whole-step graph depend on which hardware queues the graph's internal streams land on?
HIP hands hardware queues to streams round-robin; the graph's parallel branches run on streams the runtime creates at instantiation.
k dummy streams created (and used once) before the k-th capture shift that assignment.  Usage: graph_queue_probe.py [fp32|bf16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(0)
solver = AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard", image_ch=1, num_classes=4, learning_rate=1e-4, use_gpu=True,
                                               compute_dtype=dtype)
IMG, SEG, _ = bench.MASKS["targeted" if dtype == "bf16" else "dropout"]
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, device)
for _ in range(5):
    solver.cooperative_step(clean, label, noisy, IMG, SEG)
torch.cuda.synchronize()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


print(f"{dtype} eager two chains: {timed(lambda: solver.cooperative_step(clean, label, noisy, IMG, SEG)):.3f} ms", flush=True)
dummies = []
for k in range(6):
    g = CooperativeStepGraph(solver, IMG, SEG)
    g(clean, label, noisy)
    print(f"{dtype} graph captured behind {len(dummies)} dummy streams: {timed(lambda: g(clean, label, noisy)):.3f} ms", flush=True)
    del g
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.zeros(8, device=device).add_(1)
    dummies.append(s)
    torch.cuda.synchronize()
