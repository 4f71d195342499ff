#!/usr/bin/env python3
"""Weight gradients of the step's layer shapes (n = 16 / 32, BatchNorm groups 1 / 2, prologue, virtual output gradient, up-sampled input) saved to a
file; run once per kernel variant (CTL_TOOL_LIB / CTL_X3W_* hooks) and compare with `cmp A B`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        da, db = a[k]
        ea, eb = b[k]
        sw = float(ea.abs().max())
        print(f"{k}: max|dw| {sw:.3e}  rel diff dw {float((da - ea).abs().max()) / max(sw, 1e-30):.2e}  db {float((db - eb).abs().max()) / max(float(eb.abs().max()), 1e-30):.2e}  finite {bool(torch.isfinite(da).all())} {bool(torch.isfinite(ea).all())}")
    sys.exit(0)
import _variant
_ffi = _variant.use_variant()
from cooperative_training_and_latent_space_data_augmentation_amd import ops
SHAPES = [(16, 64, 64, 64, 1, 0), (32, 64, 64, 64, 2, 0), (32, 128, 128, 32, 2, 0), (32, 128, 128, 16, 2, 0), (32, 32, 32, 128, 2, 0), (16, 128, 64, 32, 1, 1), (32, 128, 64, 32, 2, 1),
          (32, 64, 32, 64, 2, 1), (16, 32, 32, 128, 1, 0)]
out = {}
for n, cin, cout, h, groups, up in SHAPES:
    g = torch.Generator().manual_seed(n + cin + h)
    ho = 2 * h if up else h
    x = torch.randn(n, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    gt = torch.randn(n, cout, ho, ho, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    u = torch.randn(n, cout, ho, ho, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    coef = (torch.randn(groups, 3, cout, generator=g) * 0.5).cuda()
    sc, sh = (torch.rand(groups, cin, generator=g) + 0.5).cuda(), (torch.randn(groups, cin, generator=g) * 0.3).cuda()
    for two in (0, 1):
        for pro in (0, 1):
            d = _ffi.conv_desc(n=n, hin=h, win=h, cin=cin, hout=ho, wout=ho, cout=cout, ks=3, in_mode=_ffi.IN_UP2 if up else 0, pro_affine=pro, pro_slope=0.2, groups=groups, dt=_ffi.DT_X3)
            dw, db = torch.zeros(cout, cin, 3, 3, device="cuda"), torch.zeros(cout, device="cuda")
            kw = dict(dy2=u, dy_coef=coef) if two else {}
            if pro:
                kw.update(pro_scale=sc, pro_shift=sh)
            for rep in range(3):          # (repeat: a race shows up as run-to-run differences)
                ops.conv_wgrad(d, x, gt, dw, (cin * 9, 9, 3, 1), dbias=db, **kw)
                torch.cuda.synchronize()
                if rep == 0:
                    first = dw.clone()
                elif not torch.equal(first, dw):
                    print(f"NOT DETERMINISTIC: n{n} {cin}->{cout} @{ho} groups {groups} up {up} dy2 {two} pro {pro}: {int((first != dw).sum())} elements differ")
            out[f"n{n} {cin}->{cout} @{ho} g{groups} up{up} dy2={two} pro={pro}"] = (dw.cpu(), db.cpu())
torch.save(out, sys.argv[1])
print("saved", sys.argv[1], len(out))
