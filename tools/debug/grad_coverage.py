"""Does a mode-A backward plan write EVERY parameter gradient?  The flat buffer is pre-filled with NaN and the memset op suppressed; any
parameter the plan does not write shows up (none does).  The memset in front of the plan stays all the same: the flat buffer has alignment
padding between the parameters, which the data-parallel all-reduce and the flat Adam launch read (dropping it: tests/test_dist_gpu.py
fails on uninitialised padding; 6 launches and 0.005 ms per step were at stake).   python tools/debug/grad_coverage.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import nets
NET_INPUT = {"image_encoder": (1, 64, 64), "shape_encoder": (4, 64, 64), "segmentation_decoder": (128, 4, 4), "shape_decoder": (128, 4, 4), "image_decoder": (128, 4, 4)}
for dtype in ("fp32", "bf16"):
    for name, (c, h, w) in NET_INPUT.items():
        res = {}
        for skip in (False, True):
            torch.manual_seed(0)
            net = nets.build_networks(device="cuda", dtype=dtype)[name]
            net.train()
            x = torch.rand(4, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
            orig_zero, orig_empty = nets.PlanBuilder.zero, torch.empty
            if skip:
                nets.PlanBuilder.zero = lambda self, ref, nbytes: None if ref[0] == nets.S_GRAD else orig_zero(self, ref, nbytes)
                def nan_empty(*a, **k):
                    t = orig_empty(*a, **k)
                    if t.dtype == torch.float32 and t.dim() == 1 and t.numel() == net._pcount: t.fill_(float("nan"))
                    return t
                torch.empty = nan_empty
            try:
                y = net(x)
                y = y if isinstance(y, tuple) else (y,)
                torch.manual_seed(1)
                torch.autograd.backward(y, [torch.randn_like(t) for t in y])
                torch.cuda.synchronize()
            finally:
                nets.PlanBuilder.zero, torch.empty = orig_zero, orig_empty
            res[skip] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        bad = [n for n, g in res[True].items() if not torch.isfinite(g).all() or not torch.equal(g, res[False][n])]
        print(f"{dtype} {name}: {len(res[True])} parameters, not written by the plan: {bad if bad else 'none'}")
