"""GPU helper for rocprofv3 --pmc runs: one conv layer of the bs16 workload, forward (with BN statistics) and wgrad.
   python tools/single_conv.py <cin> <cout> <h> <ks> <stride> <in_mode> [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check
cin, cout, h, ks, stride, mode = [int(a) for a in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
n = 16
hv = h * (2 if mode else 1)
ho = (hv + 1) // 2 if stride == 2 else hv
x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.1
b = torch.zeros(cout, device="cuda")
d = _ffi.conv_desc(n=n, hin=h, win=h, cin=cin, hout=ho, wout=ho, cout=cout, ks=ks, stride=stride, in_mode=mode,
                   epi_flags=_ffi.EPI_BIAS | _ffi.EPI_STATS)
dp = _ffi.desc_ptr(d)
wp = ops.pack_oihw_fwd(w)
y = torch.empty(n, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
st = torch.empty(lib.ctl_conv_stats_floats(dp), device="cuda")
dy = torch.randn(n, cout, ho, ho, device="cuda").contiguous(memory_format=torch.channels_last)
wpart = torch.empty(lib.ctl_wgrad_partial_floats(dp), device="cuda"); bpart = torch.empty(lib.ctl_wgrad_bias_partial_floats(dp), device="cuda")
for _ in range(iters):
    check(lib.ctl_conv_forward(dp, x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, None, None, None, y.data_ptr(), st.data_ptr(), ops.stream_ptr()))
    if mode != 2:
        check(lib.ctl_conv_wgrad(dp, x.data_ptr(), None, None, dy.data_ptr(), wpart.data_ptr(), bpart.data_ptr(), ops.stream_ptr()))
torch.cuda.synchronize()
