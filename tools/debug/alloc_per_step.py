"""Debug: device allocations (hipMalloc calls of the caching allocator) and reserved bytes per eager step -- does the pool reach a steady state?
   python tools/debug/alloc_per_step.py [bf16|fp32] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype=dt)
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, torch.device("cuda"))
ci, cs = (bench.TGT_IMG, bench.TGT_SEG) if dt == "bf16" else (bench.DROP_IMG, bench.DROP_SEG)
prev = 0
for i in range(steps):
    s.cooperative_step(clean, label, noisy, ci, cs)
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    n = st.get("num_device_alloc", 0)
    if n != prev or i % 10 == 0:
        print(f"step {i:3d}: device allocs {n:4d} (+{n - prev})  reserved {st['reserved_bytes.all.current'] / 2**20:9.1f} MiB  "
              f"allocated {st['allocated_bytes.all.current'] / 2**20:9.1f} MiB  active peak {st['active_bytes.all.peak'] / 2**20:9.1f} MiB  frees {st.get('num_device_free', 0)}")
    prev = n
