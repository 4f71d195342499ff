"""GPU tuning helper for the bf16 family (BASELINE config 3): single layers of the bs16 256^2 workload with bf16-stored tensors,
timed with events over 30 launches; prints microseconds, algorithmic GB/s and the fraction of the 8 TB/s HBM peak.
   python tools/bench_conv16.py            (CTL_TOOL_LIB selects an A/B build; CTL_PERSIST needs a -DCTL_TUNING one)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import use_variant
_ffi = use_variant()                    # CTL_TOOL_LIB=<variant> selects an A/B build (tools/build_variant.sh)
from cooperative_training_and_latent_space_data_augmentation_amd import ops
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check

BF = _ffi.DT_BF16 | _ffi.DT_X16 | _ffi.DT_Y16
N = 16
LAYERS = [  # name, cin, cout, h(in), ks, stride, in_mode, prologue, stats
    ("c16-16@256 plain+stats", 16, 16, 256, 3, 1, 0, False, True), ("c16-16@256 pro+stats", 16, 16, 256, 3, 1, 0, True, True),
    ("c16-16@256 dgrad", 16, 16, 256, 3, 1, 0, False, False), ("c32-32@128 pro+stats", 32, 32, 128, 3, 1, 0, True, True),
    ("c64-64@64 pro+stats", 64, 64, 64, 3, 1, 0, True, True), ("c128-128@32 pro+stats", 128, 128, 32, 3, 1, 0, True, True),
    ("1x1 16-16@256", 16, 16, 256, 1, 1, 0, False, False), ("up 32-16@128", 32, 16, 128, 3, 1, 1, False, True),
    ("s2 16-32@256", 16, 32, 256, 3, 2, 0, False, False),
]


def timed(run, iters=30):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    res = {}
    for name, cin, cout, h, ks, stride, mode, pro, stats in LAYERS:
        hv = h * (2 if mode else 1)
        ho = (hv + 1) // 2 if stride == 2 else hv
        x = torch.randn(N, cin, h, h, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.1
        wp = ops.pack_oihw_fwd_bf16(w)
        b = torch.zeros(cout, device="cuda")
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        d = _ffi.conv_desc(n=N, hin=h, win=h, cin=cin, hout=ho, wout=ho, cout=cout, ks=ks, stride=stride, in_mode=mode,
                           epi_flags=_ffi.EPI_BIAS | (_ffi.EPI_STATS if stats else 0), pro_affine=1 if pro else 0, pro_slope=0.2, dt=BF)
        y = torch.empty(N, cout, ho, ho, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        st = torch.empty(max(lib.ctl_conv_stats_floats(_ffi.desc_ptr(d)), 1), device="cuda")
        dp = _ffi.desc_ptr(d)
        run = lambda: check(lib.ctl_conv_forward(dp, x.data_ptr(), wp.data_ptr(), b.data_ptr(), sc.data_ptr() if pro else None,
                                                 sh.data_ptr() if pro else None, None, None, None, y.data_ptr(),
                                                 st.data_ptr() if stats else None, ops.stream_ptr()))
        has_tm = hasattr(lib, "ctl_debug_timing16")           # -DCTL_TIMING16 variant build
        import ctypes as C
        tm = (C.c_ulonglong * 12)()
        if has_tm:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            lib.ctl_debug_timing16(tm)                         # reset
        us = timed(run)
        nbytes = 2.0 * (x.numel() + y.numel())
        res[name] = {"us": round(us, 1), "GBs": round(nbytes / us / 1e3), "hbm_frac": round(nbytes / us / 1e3 / 8000, 3)}
        if has_tm:
            lib.ctl_debug_timing16(tm)
            steps = max(tm[6], 1)
            names = ["issue", "mfma", "bar_rd", "stage", "bar_wr", "epi"]
            res[name]["phase_cycles_per_step"] = {k: round(tm[i] / steps) for i, k in enumerate(names)}
            res[name]["setup_per_step"] = round(tm[7] / steps)
            res[name]["steps_per_wave"] = round(steps / max(tm[11], 1), 2)
            mhz = 100.0 * tm[8] / max(tm[9], 1)
            res[name]["memtime_MHz"] = round(mhz, 1)
            res[name]["wave_span_us"] = {"mean": round(tm[9] / max(tm[11], 1) / 30 / 100.0, 1), "max": round(tm[10] / 30 / mhz, 1)}
    # the BatchNorm-backward element-wise passes on bf16 tensors (16 ch, 256^2): reduce (2 reads), apply (2 reads + 1 write)
    c, hw = 16, 256 * 256
    dy = torch.randn(N, c, 256, 256, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    u = torch.randn_like(dy)
    dx = torch.empty_like(dy)
    sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1
    part = torch.empty(_ffi.RED_BLOCKS * 2 * c, device="cuda")
    coef = torch.randn(3 * c, device="cuda")
    us = timed(lambda: check(lib.ctl_bwd_reduce_dt(1, dy.data_ptr(), None, u.data_ptr(), sc.data_ptr(), sh.data_ptr(), 0.2, N * hw, c,
                                                   part.data_ptr(), 1, 1 | 4, None, ops.stream_ptr())))
    nb = 2.0 * 2 * dy.numel()
    res["bwd_reduce<1> 16ch@256"] = {"us": round(us, 1), "GBs": round(nb / us / 1e3), "hbm_frac": round(nb / us / 1e3 / 8000, 3)}
    us = timed(lambda: check(lib.ctl_bwd_apply_dt(1, dy.data_ptr(), None, u.data_ptr(), sc.data_ptr(), sh.data_ptr(), 0.2, coef.data_ptr(),
                                                  N * hw, c, None, dx.data_ptr(), 1, 1 | 4 | 16, ops.stream_ptr())))
    nb = 2.0 * 3 * dy.numel()
    res["bwd_apply<1> 16ch@256"] = {"us": round(us, 1), "GBs": round(nb / us / 1e3), "hbm_frac": round(nb / us / 1e3 / 8000, 3)}
    # weight gradients (bf16 x and dy): the kernel alone, without the split reduction
    for name, cin, cout, h, ks in (("wgrad 3x3 16-16@256", 16, 16, 256, 3), ("wgrad 3x3 32-32@128", 32, 32, 128, 3), ("wgrad 1x1 16-16@256", 16, 16, 256, 1)):
        x = torch.randn(N, cin, h, h, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dyw = torch.randn(N, cout, h, h, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        d = _ffi.conv_desc(n=N, hin=h, win=h, cin=cin, hout=h, wout=h, cout=cout, ks=ks, dt=BF)
        dp = _ffi.desc_ptr(d)
        wpart = torch.empty(lib.ctl_wgrad_partial_floats(dp), device="cuda")
        bpart = torch.empty(lib.ctl_wgrad_bias_partial_floats(dp), device="cuda")
        us = timed(lambda: check(lib.ctl_conv_wgrad(dp, x.data_ptr(), None, None, dyw.data_ptr(), wpart.data_ptr(), bpart.data_ptr(), ops.stream_ptr())))
        nb = 2.0 * (x.numel() + dyw.numel())
        res[name] = {"us": round(us, 1), "GBs": round(nb / us / 1e3), "hbm_frac": round(nb / us / 1e3 / 8000, 3), "splits": lib.ctl_wgrad_splits(dp)}
    for k, v in res.items():
        print(f"  {k:28s} {v['us']:8.1f} us  {v['GBs']:6d} GB/s  {v['hbm_frac']:.3f}", v.get("phase_cycles_per_step", ""), v.get("setup_per_step", ""), v.get("steps_per_wave", ""), v.get("memtime_MHz", ""), v.get("wave_span_us", ""), v.get("splits", ""))
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
