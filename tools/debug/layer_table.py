"""Per layer SHAPE: launches, time and efficiency inside one cooperative step (single stream, every launch bracketed).  Needs a -DCTL_TUNING
build (CTL_PROF_SHAPES):  CTL_TOOL_LIB=tuning CTL_PROF_SHAPES=1 python tools/debug/layer_table.py [bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _variant import use_variant
_ffi = use_variant(os.environ.get("CTL_TOOL_LIB", "tuning"))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench
dt = "bf16" if "bf16" in sys.argv else "fp32"
cfg = (bench.TGT_IMG, bench.TGT_SEG) if dt == "bf16" else (bench.DROP_IMG, bench.DROP_SEG)
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype=dt)
s.two_streams = False
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, torch.device("cuda"))
for _ in range(4): s.cooperative_step(clean, label, noisy, *cfg)
torch.cuda.synchronize()
_ffi.prof_start("")
N = 4
for _ in range(N): s.cooperative_step(clean, label, noisy, *cfg)
torch.cuda.synchronize()
prof = _ffi.prof_stop()
peak = 2500.0 if dt == "bf16" else 157.3
rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in prof.values()) / N
print(f"{dt}: bracketed kernel time {tot:.3f} ms/step")
for k, v in rows[:70]:
    ms = v["ms"] / N
    tf, gb = v["flops"] / v["ms"] / 1e9, v["bytes"] / v["ms"] / 1e6
    print(f"{k:78s} {v['launches']/N:5.1f}/step {ms:7.3f} ms  avg {1e3*v['ms']/v['launches']:6.1f} us  {tf:6.1f} TF ({tf/peak:.2f})  {gb:6.0f} GB/s ({gb/8000:.2f})")
