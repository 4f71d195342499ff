"""Debug: where does the HOST spend its time issuing one eager cooperative step?  (cProfile over 20 steps; GPU work is asynchronous)
   python tools/debug/host_profile.py [bf16|fp32]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype=dt)
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256     # a small size (32) makes the step host-bound: step time = host time
clean, label, noisy, _ = bench.synthetic(16, size, size, 1000, torch.device("cuda"))
ci, cs = (bench.TGT_IMG, bench.TGT_SEG) if dt == "bf16" else (bench.DROP_IMG, bench.DROP_SEG)
for _ in range(8):
    s.cooperative_step(clean, label, noisy, ci, cs)
torch.cuda.synchronize()
t = time.perf_counter()
issue = []
for _ in range(20):
    t0 = time.perf_counter()
    s.cooperative_step(clean, label, noisy, ci, cs)
    issue.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"{dt}: {1e3 * (time.perf_counter() - t) / 20:.2f} ms per step, host issue median {1e3 * sorted(issue)[10]:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    s.cooperative_step(clean, label, noisy, ci, cs)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
