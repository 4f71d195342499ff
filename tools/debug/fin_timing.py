"""Phase timers of the forward BatchNorm finalize inside the real step (variant build: tools/build_variant.sh tmfin -DCTL_TIMING_FIN ctl_elem.hip;
CTL_TOOL_LIB=tmfin python tools/debug/fin_timing.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _variant
_variant.use_variant()
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)
clean = torch.rand(16, 1, 256, 256, device="cuda"); noisy = (clean + 0.1 * torch.randn_like(clean)).clamp(0, 1)
label = torch.randint(0, 4, (16, 256, 256), device="cuda")
for _ in range(5): s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
_ffi.lib.ctl_debug_timing_fin(out)
for _ in range(10): s.cooperative_step(clean, label, noisy, bench.DROP_IMG, bench.DROP_SEG)
torch.cuda.synchronize()
_ffi.lib.ctl_debug_timing_fin(out)
n = max(out[3], 1)
print("forward finalize launches %d (per step %.0f): rows %.0f groups %.2f | cycles per launch (thread 0 of block 0): row loads %.0f  block sum %.0f  coefficients + stores %.0f | "
      "first to last instruction %.2f us" % (out[3], out[3] / 10, out[4] / n, out[5] / n, out[0] / n, out[1] / n, out[2] / n, out[6] / n / 100.0))
