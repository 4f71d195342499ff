#!/usr/bin/env python3
"""X3 weight gradients per layer (HIP events, back-to-back launches): ctl_conv_wgrad_ex on the step's 3x3 stride-1 layer shapes, plain and with the
virtual output gradient (dy2).  CTL_TOOL_LIB=tuning CTL_X3W_PC=0 times the single-role kernel (tools/build_variant.sh tuning -DCTL_TUNING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _variant
_ffi = _variant.use_variant()
from cooperative_training_and_latent_space_data_augmentation_amd import ops
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check

LAYERS = [(16, 32, 32, 128, 0), (16, 64, 64, 64, 0), (16, 128, 128, 32, 0), (32, 64, 64, 64, 0), (32, 128, 128, 32, 0), (16, 128, 64, 32, 1), (16, 64, 32, 64, 1),
          (32, 32, 32, 128, 0), (16, 128, 128, 16, 0), (32, 128, 128, 16, 0), (16, 16, 16, 256, 0), (16, 32, 16, 128, 0)]


def main():
    for n, cin, cout, h, up in LAYERS:
        ho = 2 * h if up else h
        x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
        u = torch.randn(n, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
        coef = torch.randn(1, 3, cout, device="cuda")
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
        d = _ffi.conv_desc(n=n, hin=h, win=h, cin=cin, hout=ho, wout=ho, cout=cout, ks=3, in_mode=_ffi.IN_UP2 if up else 0, pro_affine=1, pro_slope=0.2, dt=_ffi.DT_X3)
        dp = _ffi.desc_ptr(d)
        wpart = torch.empty(lib.ctl_wgrad_partial_floats(dp), device="cuda")
        bpart = torch.empty(lib.ctl_wgrad_bias_partial_floats(dp), device="cuda")
        out = []
        for two in (False, True):
            run = lambda: check(lib.ctl_conv_wgrad_ex(dp, x.data_ptr(), sc.data_ptr(), sh.data_ptr(), dy.data_ptr(), u.data_ptr() if two else None,
                                                      coef.data_ptr() if two else None, wpart.data_ptr(), bpart.data_ptr(), ops.stream_ptr()))
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            if hasattr(lib._lib if lib._lib else lib.load(), "ctl_debug_timing_x3w"):
                import ctypes
                lib.ctl_debug_timing_x3w((ctypes.c_ulonglong * 10)())      # reset
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / 30)
            if hasattr(lib._lib if lib._lib else lib.load(), "ctl_debug_timing_x3w"):
                import ctypes
                tm = (ctypes.c_ulonglong * 10)()
                lib.ctl_debug_timing_x3w(tm)
                if tm[5]:
                    t = float(tm[5])
                    blocks = 35.0 * 256
                    print("     per tile and wave: producer stage %d issue %d barrier %d | consumer barrier %d mfma %d | per block and wave: cons head %d tail %d  prod head %d  span %d  tiles/block %.1f"
                          % (tm[0] / t, tm[1] / t, tm[2] / t, tm[3] / t, tm[4] / t, tm[6] / (4 * blocks), tm[7] / (4 * blocks), tm[8] / (4 * blocks), tm[9] / (4 * blocks), t / (4 * blocks)))
        fl = 2.0 * n * ho * ho * cin * cout * 9
        print(f"n{n} {cin:3d}->{cout:3d} @{ho}x{ho} up={up} splits {lib.ctl_wgrad_splits(dp):3d}: plain {out[0]:6.1f} us ({fl / out[0] / 1e6:6.1f} TF alg = {fl / out[0] / 1e6 / 416.7:.3f})   "
              f"dy2 {out[1]:6.1f} us ({fl / out[1] / 1e6:6.1f} TF alg = {fl / out[1] / 1e6 / 416.7:.3f})", flush=True)


if __name__ == "__main__":
    main()
