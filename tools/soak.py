"""Soak run: N cooperative steps at the bench size (16 x 256 x 256) with the mask scheme changing every step (dropout / channel /
spatial, both loss pairings) and fresh synthetic batches; every 100 steps the losses and all parameters are checked for finiteness.
   python tools/soak.py [steps=3000] [fp32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype=sys.argv[2] if len(sys.argv) > 2 else "fp32")
def cfg(loss, kind): return {"loss_name": loss, "mask_type": kind, "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
schemes = [(bench.DROP_IMG, bench.DROP_SEG), (cfg("mse", "channel"), cfg("ce", "spatial")), (cfg("mse", "spatial"), cfg("ce", "channel")),
           (cfg("mse", "channel"), cfg("ce", "channel"))]
g = torch.Generator(device="cuda").manual_seed(1)
t0 = time.time()
for it in range(steps):
    clean = torch.rand(16, 1, 256, 256, device="cuda", generator=g)
    label = torch.randint(0, 4, (16, 256, 256), device="cuda", generator=g)
    noisy = (clean + 0.05 * torch.randn(clean.shape, device="cuda", generator=g)).clamp(0, 1)
    losses = s.cooperative_step(clean, label, noisy, *schemes[it % len(schemes)])
    if it % 100 == 99:
        v = [float(x) for x in losses]
        ok = all(x == x and abs(x) < 1e4 for x in v) and all(bool(torch.isfinite(m._flat_data).all()) for m in s.model.values())
        print(it + 1, f"{time.time() - t0:.0f}s", " ".join(f"{x:.3f}" for x in v), "OK" if ok else "NOT FINITE", flush=True)
        assert ok
print("soak done:", steps, "steps")
