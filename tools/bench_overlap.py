#!/usr/bin/env python3
"""Does a layer's weight gradient overlap with its data gradient when they are issued on two streams?  (They are independent: both read dy.)
Captured into a HIP graph (20 pairs) so that the host's launch rate does not matter: sequential on one stream vs forked on two."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops

DEV = "cuda"
dev = lambda x: x.to(DEV).contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x.to(DEV).contiguous()


def graph_time(fn, reps=20):
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(side)
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn(side)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def one(n, cin, cout, h, w, dt):
    g = torch.Generator().manual_seed(1)
    x, dy = dev(torch.randn(n, cin, h, w, generator=g)), dev(torch.randn(n, cout, h, w, generator=g))
    wt = dev(torch.randn(cout, cin, 3, 3, generator=g) * 0.2)
    wp = (ops.pack_oihw_dgrad_x3 if dt else ops.pack_oihw_dgrad)(wt)
    dd = _ffi.conv_desc(n=n, hin=h, win=w, cin=cout, hout=h, wout=w, cout=cin, ks=3, dt=dt)          # data gradient: cout -> cin
    dwd = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, dt=dt)         # weight gradient of cin -> cout
    dx = torch.empty_like(x)
    wpart = torch.empty(_ffi.lib.ctl_wgrad_partial_floats(_ffi.desc_ptr(dwd)), device=DEV)
    other = torch.cuda.Stream()

    def dgrad(s):
        _ffi.check(_ffi.lib.ctl_conv_forward_ex(_ffi.desc_ptr(dd), dy.data_ptr(), wp.data_ptr(), None, None, None, None, None, None, None, None, dx.data_ptr(), None, None, None, s.cuda_stream))

    def wgrad(s):
        _ffi.check(_ffi.lib.ctl_conv_wgrad_ex(_ffi.desc_ptr(dwd), x.data_ptr(), None, None, dy.data_ptr(), None, None, wpart.data_ptr(), None, s.cuda_stream))

    def seq(s):
        dgrad(s); wgrad(s)

    def forked(s):
        other.wait_stream(s)
        wgrad(other)
        dgrad(s)
        s.wait_stream(other)

    td, tw = graph_time(lambda s: dgrad(s)), graph_time(lambda s: wgrad(s))
    ts, tf = graph_time(seq), graph_time(forked)
    print(f"n{n} {cin}->{cout} @{h}x{w} dt={dt}: dgrad {td:6.1f} us  wgrad {tw:6.1f} us  sequential {ts:6.1f} us  forked {tf:6.1f} us  ({ts / tf:.2f}x)", flush=True)


if __name__ == "__main__":
    for dt in (_ffi.DT_X3, 0):
        for args in [(16, 16, 16, 256, 256), (16, 32, 32, 128, 128), (16, 64, 64, 64, 64), (16, 128, 128, 32, 32)]:
            one(*args, dt)
