"""Per kernel template of the conv family: launches, time, algorithmic TFLOP/s and TB/s inside the bench workload (in-process
HIP-event profiler; run on the GPU box).  Shows which shapes sit far from the MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess, json
os.environ.setdefault("CTL_PROF_ALL", "1")
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)
clean = torch.rand(16, 1, 256, 256, device="cuda"); noisy = (clean + 0.1 * torch.randn_like(clean)).clamp(0, 1)
label = torch.randint(0, 4, (16, 256, 256), device="cuda")
for _ in range(3): s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
torch.cuda.synchronize()
_ffi.prof_start("")
N = 5
for _ in range(N): s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
torch.cuda.synchronize()
p = _ffi.prof_stop()
rows = sorted(p.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows) / N
fl = sum(v["flops"] for _, v in rows) / N
print(f"conv family: {tot:.2f} ms/step, {fl/1e9:.1f} GFLOP/step -> {fl/tot/1e9:.1f} TFLOP/s average; MFMA floor {fl/157.3e9:.2f} ms")
print(f"{'kernel':44s} {'calls/step':>10s} {'ms/step':>8s} {'us/call':>8s} {'TFLOP/s':>8s} {'TB/s':>6s}")
for k, v in rows:
    ms = v["ms"] / N
    print(f"{k:44s} {v['launches']/N:10.0f} {ms:8.3f} {1e3*v['ms']/v['launches']:8.1f} {v['flops']/v['ms']/1e9:8.1f} {v['bytes']/v['ms']/1e9:6.2f}")
