"""GPU tuning helper: time single conv layers of the bs16 workload under forced tile configurations.
   python tools/bench_conv.py            (each layer x each valid "mt,tw,nt")"""
import os, sys, subprocess, json
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAYERS = [  # name, cin, cout, h(in), ks, stride, in_mode
    ("c16-16@256", 16, 16, 256, 3, 1, 0), ("c32-32@128", 32, 32, 128, 3, 1, 0), ("c16-32@128", 16, 32, 128, 3, 1, 0),
    ("c64-64@64", 64, 64, 64, 3, 1, 0), ("c128-128@32", 128, 128, 32, 3, 1, 0), ("c128-128@16", 128, 128, 16, 3, 1, 0),
    ("c128-64@32", 128, 64, 32, 3, 1, 0), ("up16-16@128", 16, 16, 128, 3, 1, 1), ("1x1 16-16@256", 16, 16, 256, 1, 1, 0),
    ("s2 16-16@256", 16, 16, 256, 3, 2, 0), ("zins 128@16", 128, 128, 16, 3, 1, 2),
    # round 3: the mid-resolution layers of the n=16 passes (layer table: 0.34-0.5 of peak inside the step)
    ("c16-16@128", 16, 16, 128, 3, 1, 0), ("c32-32@64", 32, 32, 64, 3, 1, 0), ("c64-64@32", 64, 64, 32, 3, 1, 0), ("c32-16@128", 32, 16, 128, 3, 1, 0),
    ("c64-32@64", 64, 32, 64, 3, 1, 0), ("1x1 128-64@32", 128, 64, 32, 1, 1, 0), ("1x1 32-32@64", 32, 32, 64, 1, 1, 0),
    # round 4: the stride-2 3x3 convs of the down-sampling blocks
    ("s2 16-32@256", 16, 32, 256, 3, 2, 0), ("s2 32-64@128", 32, 64, 128, 3, 2, 0), ("s2 64-128@64", 64, 128, 64, 3, 2, 0), ("s2 32-32@128", 32, 32, 128, 3, 2, 0),
]
CFGS = ["4,32,1", "2,16,1", "1,16,1", "4,32,2", "2,16,2", "1,16,2", "4,32,4", "2,16,4", "1,16,4"]

def child(kind):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _variant import use_variant
    _ffi = use_variant()                # CTL_TOOL_LIB=tuning: the per-config sweep below needs the CTL_FORCE_CFG hook of a -DCTL_TUNING build
    from cooperative_training_and_latent_space_data_augmentation_amd import ops
    from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check
    res = {}
    x3 = bool(os.environ.get("CTL_BENCH_X3"))          # the same layers as CTL_DT_X3 launches (those X3 covers)
    for name, cin, cout, h, ks, stride, mode in LAYERS:
        n = int(os.environ.get("CTL_BENCH_N", "16"))
        if x3 and (ks == 1 or cin % 16 or cout % 16):
            continue
        hv = h * (2 if mode else 1)
        ho = (hv + 1) // 2 if stride == 2 else hv
        x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.1
        if os.environ.get("CTL_ZERO_DATA"):      # power/DVFS probe: all-zero operands
            x.zero_(); w.zero_()
        d = _ffi.conv_desc(n=n, hin=h, win=h, cin=cin, hout=ho, wout=ho, cout=cout, ks=ks, stride=stride, in_mode=mode,
                           epi_flags=_ffi.EPI_BIAS | (_ffi.EPI_STATS if kind == "fwd" else 0), dt=_ffi.DT_X3 if x3 else 0)
        b = torch.zeros(cout, device="cuda")
        flops = 2.0 * n * ho * ho * cout * cin * ks * ks
        if kind == "fwd":
            wp = ops.pack_oihw_fwd_x3(w) if x3 else ops.pack_oihw_fwd(w)
            y = torch.empty(n, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
            st = torch.empty(lib.ctl_conv_stats_floats(_ffi.desc_ptr(d)), device="cuda")
            run = lambda: check(lib.ctl_conv_forward(_ffi.desc_ptr(d), x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, None, None,
                                                     None, y.data_ptr(), st.data_ptr(), ops.stream_ptr()))
        else:
            if mode == 2: continue
            dy = torch.randn(n, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
            dp = _ffi.desc_ptr(d)
            wpart = torch.empty(lib.ctl_wgrad_partial_floats(dp), device="cuda"); bpart = torch.empty(lib.ctl_wgrad_bias_partial_floats(dp), device="cuda")
            run = lambda: check(lib.ctl_conv_wgrad(dp, x.data_ptr(), None, None, dy.data_ptr(), wpart.data_ptr(), bpart.data_ptr(), ops.stream_ptr()))
        try:
            for _ in range(3): run()
        except Exception as e:
            res[name] = None; continue
        torch.cuda.synchronize()
        tm = (C.c_ulonglong * 12)()
        has_tm = hasattr(lib, "ctl_debug_timing")      # -DCTL_TIMING variant build (tools/build_variant.sh)
        if has_tm: lib.ctl_debug_timing(tm)                               # reset
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res[name] = (round(us, 1), round(flops / us / 1e6, 1))
        if has_tm:
            lib.ctl_debug_timing(tm)
            steps = max(tm[6], 1)          # (tile, chunk) steps summed over waves
            names = ["issue", "mfma", "bar_rd", "stage", "bar_wr", "epi" if kind == "fwd" else "tail(total)"]
            res[name] = res[name] + ({k: round(tm[i] / steps) for i, k in enumerate(names)}, {"setup/step": round(tm[7] / steps, 1), "steps/launch": steps // 20, "memtime_MHz": round(100.0 * tm[8] / max(tm[9], 1), 1), "span_mean_us": round(tm[9] / max(tm[11], 1) / 20 / 100.0, 1), "span_max_us": round(tm[10] / 20 / (100.0 * tm[8] / max(tm[9], 1)), 1)})
    print("RESULT " + json.dumps(res))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        for kind in ("fwd", "wgrad"):
            table = {}
            for cfg in ["auto"] + CFGS:
                env = dict(os.environ)
                if cfg != "auto": env["CTL_FORCE_CFG"] = cfg
                out = subprocess.run([sys.executable, __file__, "child", kind], env=env, capture_output=True, text=True).stdout
                line = [l for l in out.splitlines() if l.startswith("RESULT ")]
                table[cfg] = json.loads(line[0][7:]) if line else {}
            print(f"==== {kind}: microseconds (TFLOP/s) per config")
            print(f"{'layer':16s}" + "".join(f"{c:>16s}" for c in table))
            for name, *_ in LAYERS:
                print(f"{name:16s}" + "".join(f"{(str(t.get(name)[0]) + ' (' + str(t.get(name)[1]) + ')') if t.get(name) else '-':>16s}" for t in table.values()))
