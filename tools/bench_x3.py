#!/usr/bin/env python3
"""X3 prototype gate: the fp32 conv on the bf16 matrix pipe by an exact three-way operand split, against the fp32-MFMA
kernel on the same descriptors: error of both against an fp64 CPU reference, and time per launch (HIP events, back-to-back launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import _variant
_variant.use_variant()
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops

DEV = "cuda"
dev = lambda x: x.to(DEV).contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x.to(DEV).contiguous()


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def one(n, cin, cout, h, w, check_n=2, pro=False, stats=True):
    g = torch.Generator().manual_seed(cin * 7 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    flags = _ffi.EPI_STATS if stats else 0
    kw = dict(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, epi_flags=flags)
    if pro:
        kw.update(pro_affine=1, pro_slope=0.2)
    d0, d3 = _ffi.conv_desc(**kw), _ffi.conv_desc(dt=_ffi.DT_X3, **kw)
    xd, w0, w3 = dev(x), ops.pack_oihw_fwd(dev(wt)), ops.pack_oihw_fwd_x3(dev(wt))
    pk = dict(pro_scale=dev(sc), pro_shift=dev(sh)) if pro else {}
    y0, _ = ops.conv_forward(d0, xd, w0, want_stats=stats, **pk)
    y3, _ = ops.conv_forward(d3, xd, w3, want_stats=stats, **pk)
    xin = x[:check_n].double()
    if pro:
        t = xin * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        xin = torch.where(t > 0, t, t * 0.2)
    ref = F.conv2d(xin, wt.double(), padding=1)
    e0 = float((y0[:check_n].cpu().double() - ref).abs().max() / ref.abs().max())
    e3 = float((y3[:check_n].cpu().double() - ref).abs().max() / ref.abs().max())
    yb = torch.empty_like(y0)
    st = torch.empty(max(_ffi.lib.ctl_conv_stats_floats(_ffi.desc_ptr(d0)), _ffi.lib.ctl_conv_stats_floats(_ffi.desc_ptr(d3)), 1), device=DEV) if stats else None
    def run(d, wp):
        _ffi.check(_ffi.lib.ctl_conv_forward_ex(_ffi.desc_ptr(d), xd.data_ptr(), wp.data_ptr(), None, ops.ptr(pk.get("pro_scale")), ops.ptr(pk.get("pro_shift")),
                                                None, None, None, None, None, yb.data_ptr(), ops.ptr(st), None, None, ops.stream_ptr()))
    t0 = timeit(lambda: run(d0, w0))
    tm = None
    if hasattr(_ffi.lib._lib if _ffi.lib._lib else _ffi.lib.load(), "ctl_debug_timing_x3"):
        import ctypes
        tm = (ctypes.c_ulonglong * 12)()
        _ffi.lib.ctl_debug_timing_x3(tm)
    t3 = timeit(lambda: run(d3, w3))
    if tm is not None:
        _ffi.lib.ctl_debug_timing_x3(tm)
        steps = max(tm[6], 1)
        names = ["issue", "mfma", "bar_rd", "stage", "bar_wr", "epi"]
        print("   x3 phase cycles per (tile, chunk) step per wave:", {k: round(tm[i] / steps) for i, k in enumerate(names)}, "setup/wave", round(tm[7] / max(tm[11], 1)),
              "span/wave", round(tm[8] / max(tm[11], 1)), "MHz or vmcnt-wait/step", round(100.0 * tm[8] / max(tm[9], 1)), round(tm[9] / steps), "steps/wave", round(steps / max(tm[11], 1), 1))
    fl = 2.0 * n * h * w * cin * cout * 9
    by = 4.0 * n * h * w * (cin + cout)
    print(f"n{n} {cin:3d}->{cout:3d} @{h}x{w} pro={int(pro)}: fp32 {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF, err {e0:.2e})   x3 {t3:7.1f} us ({fl / t3 / 1e6:6.1f} TF alg, {by / t3 / 1e3:6.0f} GB/s, err {e3:.2e})   x{t0 / t3:.2f}", flush=True)


if __name__ == "__main__":
    if "--one" in sys.argv:
        one(16, 16, 16, 256, 256)
        sys.exit(0)
    for args in [(16, 16, 16, 256, 256), (16, 32, 32, 128, 128), (16, 64, 64, 64, 64), (16, 128, 128, 32, 32), (16, 128, 128, 16, 16), (32, 64, 64, 64, 64), (16, 32, 16, 128, 128)]:
        one(*args)
        one(*args, pro=True)
