"""Isolated cost of the two-tensor output gradient (dy2) in the weight-gradient kernels, fp32 and bf16:  python tools/debug/wgrad_dy2_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _variant import use_variant
use_variant()               # CTL_TOOL_LIB=<name>: an A/B build of the kernels (tools/build_variant.sh)
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check
LAYERS = [(32, 16, 16, 256), (32, 32, 32, 128), (32, 64, 64, 64), (32, 128, 128, 32), (16, 64, 64, 64), (16, 128, 128, 16)]
for b16 in ((False,) if os.environ.get("FP32_ONLY") else (False, True)):
    for n, cin, cout, h in LAYERS:
        dt = (_ffi.DT_BF16 | _ffi.DT_X16 | _ffi.DT_Y16) if b16 else 0
        td = torch.bfloat16 if b16 else torch.float32
        mk = lambda c: torch.randn(n, c, h, h, device="cuda").to(td).contiguous(memory_format=torch.channels_last)
        x, g, u = mk(cin), mk(cout), mk(cout)
        coef = torch.randn(3 * cout, device="cuda")
        d = _ffi.conv_desc(n=n, hin=h, win=h, cin=cin, hout=h, wout=h, cout=cout, ks=3, dt=dt)
        dp = _ffi.desc_ptr(d)
        wpart = torch.empty(lib.ctl_wgrad_partial_floats(dp), device="cuda"); bpart = torch.empty(lib.ctl_wgrad_bias_partial_floats(dp), device="cuda")
        res = []
        for dy2 in (None, u):
            run = lambda: check(lib.ctl_conv_wgrad_ex(dp, x.data_ptr(), None, None, g.data_ptr(), dy2.data_ptr() if dy2 is not None else None,
                                                      coef.data_ptr() if dy2 is not None else None, wpart.data_ptr(), bpart.data_ptr(), ops.stream_ptr()))
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 50)
        print(f"{'bf16' if b16 else 'fp32'} wgrad 3x3 n{n} {cin}->{cout} @{h}^2: plain {res[0]:6.1f} us   with dy2 {res[1]:6.1f} us   (+{100 * (res[1] / res[0] - 1):.1f} %)")
