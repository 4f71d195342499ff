"""How long does the CPU need to issue one training step (no GPU sync inside)?  If this approaches the step time the GPU starves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)
clean = torch.rand(16, 1, 256, 256, device="cuda"); noisy = (clean + 0.1 * torch.randn_like(clean)).clamp(0, 1)
label = torch.randint(0, 4, (16, 256, 256), device="cuda")
for _ in range(5): s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
torch.cuda.synchronize()
cpu, tot = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    cpu.append(t1 - t0); tot.append(t2 - t0)
print(f"CPU issue time per step {1e3*sum(cpu)/len(cpu):.2f} ms; step incl. GPU drain {1e3*sum(tot)/len(tot):.2f} ms (isolated steps, sync before each)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
