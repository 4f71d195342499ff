#!/usr/bin/env python3
"""Host time of one whole-step graph replay call (GPU idle before it, so no back-pressure) against the GPU time of the replay.
Usage: graph_host_cost.py [fp32|bf16] [--single-stream]"""
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
if "--single-stream" in sys.argv:
    solver.two_streams = False
IMG, SEG, _ = bench.MASKS["targeted" if dtype == "bf16" else "dropout"]
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, device)
for _ in range(5):
    solver.cooperative_step(clean, label, noisy, IMG, SEG)
g = CooperativeStepGraph(solver, IMG, SEG)
for _ in range(3):
    g(clean, label, noisy)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g(clean, label, noisy)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0)); total.append(1e3 * (t2 - t0))
host.sort(); total.sort()
print(f"{dtype} {'one chain' if not solver.two_streams else 'two chains'}: replay call returns after {host[5]:.2f} ms (median; min {host[0]:.2f}), GPU done after {total[5]:.2f} ms", flush=True)
