"""Kernel-level parity on a real MI355X: every C-ABI op against a plain PyTorch fp32 CPU reference of the same op.
Tolerance for fp32 contractions: 2e-4 * max|ref| (K up to 1152 fp32 MACs, different summation order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check  # noqa: E402

DEV = "cuda"


def dev(x):
    if x.dim() == 4:
        return x.to(DEV).contiguous(memory_format=torch.channels_last)
    return x.to(DEV).contiguous()


def close(a, b, rel=2e-4, what=""):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = rel * max(float(b.abs().max()), 1e-6) + 1e-7
    err = float((a - b).abs().max())
    assert err <= tol, f"{what}: max err {err:.3e} > tol {tol:.3e}"


def leaky(x, s):
    return torch.where(x > 0, x, x * s)


CONV_CASES = [
    # n, cin, cout, h, w
    (2, 16, 16, 32, 32), (16, 16, 16, 64, 64), (4, 16, 16, 128, 128), (2, 32, 64, 16, 16), (2, 128, 128, 8, 8),
    (2, 64, 32, 24, 20), (2, 1, 16, 32, 32), (2, 4, 16, 20, 12), (2, 16, 4, 32, 32), (3, 16, 1, 16, 16), (1, 32, 32, 6, 6),
    (2, 128, 64, 3, 3),
    # interior tiles (scalar-offset fast path) with several 16-channel chunks, ragged right/bottom edges
    (2, 32, 32, 48, 48), (1, 64, 48, 40, 72), (2, 128, 32, 36, 52), (3, 48, 16, 70, 70),
]


@pytest.mark.parametrize("n,cin,cout,h,w", CONV_CASES)
def test_conv3x3_s1_forward_bias_prologue_stats(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(n * 1000 + cin * 10 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    wp = ops.pack_oihw_fwd(dev(wt))
    # plain + bias + stats
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, epi_flags=_ffi.EPI_BIAS | _ffi.EPI_STATS)
    y, stats = ops.conv_forward(d, dev(x), wp, bias=dev(b), want_stats=True)
    ref = F.conv2d(x, wt, b, padding=1)
    close(y, ref, what="conv3x3")
    st = stats.view(-1, 2, cout).double().sum(0).cpu()
    close(st[0], ref.double().sum((0, 2, 3)), rel=1e-4, what="stats sum")
    close(st[1], (ref.double() ** 2).sum((0, 2, 3)), rel=1e-4, what="stats sumsq")
    # prologue (BN-apply + LeakyReLU) fused into staging; zero padding applies AFTER the activation
    d2 = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, epi_flags=_ffi.EPI_BIAS, pro_affine=1,
                        pro_slope=0.2)
    y2, _ = ops.conv_forward(d2, dev(x), wp, bias=dev(b), pro_scale=dev(sc), pro_shift=dev(sh))
    ref2 = F.conv2d(leaky(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2), wt, b, padding=1)
    close(y2, ref2, what="conv3x3+prologue")


@pytest.mark.parametrize("n,c,cout,h,w", [(2, 16, 16, 32, 32), (2, 32, 64, 24, 40), (3, 128, 64, 12, 8), (16, 16, 16, 128, 128), (2, 64, 32, 6, 6)])
def test_conv4x4_s2_is_pooled_3x3_data_gradient(n, c, cout, h, w):
    """4x4 stride-2 pad-1 conv (generic weights) and its use: sumpool2(conv3x3^T(dy)) == conv4x4s2(dy; K) with K summed from the
    3x3 taps (the weight-pack mode of the nearest-upsample blocks)."""
    g = torch.Generator().manual_seed(c + h)
    dy = torch.randn(n, c, h, w, generator=g)
    k4 = torch.randn(cout, c, 4, 4, generator=g) * 0.2
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h // 2, wout=w // 2, cout=cout, ks=4, stride=2)
    y, _ = ops.conv_forward(d, dev(dy), ops.pack_oihw_fwd(dev(k4)))
    close(y, F.conv2d(dy, k4, stride=2, padding=1), what="conv4x4 s2")
    # data gradient of y' = conv3x3(up2(x)) with respect to x: forward weights W [c_out_f = c][c_in_f = cout][3][3]
    wf = torch.randn(c, cout, 3, 3, generator=g) * 0.2
    x = torch.randn(n, cout, h // 2, w // 2, generator=g, requires_grad=True)
    F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wf, padding=1).backward(dy)
    K = torch.zeros(cout, c, 4, 4)
    for a in range(2):
        for b in range(2):
            for kh in range(3):
                for kw in range(3):
                    K[:, :, a + 2 - kh, b + 2 - kw] += wf[:, :, kh, kw].t()
    y2, _ = ops.conv_forward(d, dev(dy), ops.pack_oihw_fwd(dev(K)))
    close(y2, x.grad, what="pooled dgrad via 4x4 s2")


def _pack_phases(w, cout_eff, cin_eff, strides, mode):
    """Four 2x2 phase packs of a 3x3 weight through the table-driven pack kernel (modes 2 / 3 of ctl_pack_weights_batched)."""
    sub = lib.ctl_conv_wpack_floats(cin_eff, cout_eff, 2)
    table = np.asarray([[0, z * sub, cout_eff, cin_eff, 2, z, *strides, sub, mode] for z in range(4)], dtype=np.int64)
    wd, td = dev(w).contiguous(), torch.from_numpy(table).to(DEV)
    out = torch.zeros(4 * sub, device=DEV)
    check(lib.ctl_pack_weights_batched(wd.data_ptr(), out.data_ptr(), td.data_ptr(), 4, sub, ops.stream_ptr()))
    return out


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 16, 16, 16, 16), (2, 128, 64, 4, 4), (3, 32, 16, 24, 20), (16, 16, 16, 64, 64), (2, 64, 32, 40, 36)])
def test_phase_convs_upsampled_forward_and_stride2_dgrad(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + h)
    # (i) conv3x3(nearest_up(x)) == four 2x2 phase convs on x (pad code 2), with bias and BatchNorm statistics
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=2, stride=1, pad=2, nsub=4, out_h=2 * h, out_w=2 * w,
                       out_sy=2, out_sx=2, out_sub=1, epi_flags=_ffi.EPI_BIAS | _ffi.EPI_STATS)
    y = ops.empty_nhwc(n, cout, 2 * h, 2 * w, DEV)
    _, st = ops.conv_forward(d, dev(x), _pack_phases(wt, cout, cin, (cin * 9, 9, 3, 1), 2), bias=dev(b), y=y, want_stats=True)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    close(y, ref, what="phase forward of conv3x3(up2(x))")
    rows = lib.ctl_conv_stats_blocks(_ffi.desc_ptr(d))
    part = st.cpu().double().view(rows, 2, cout).sum(0)
    assert float((part[0] - ref.double().sum((0, 2, 3))).abs().max()) <= 2e-4 * float(ref.double().sum((0, 2, 3)).abs().max()) + 1e-2
    assert float((part[1] - (ref.double() ** 2).sum((0, 2, 3))).abs().max()) <= 2e-4 * float((ref.double() ** 2).sum((0, 2, 3)).max())
    # (ii) data gradient of a stride-2 pad-1 3x3 conv == four phase convs over dy (pad code 0)
    xs = torch.randn(n, cin, 2 * h, 2 * w, generator=g, requires_grad=True)
    ws = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    ys = F.conv2d(xs, ws, stride=2, padding=1)
    dy = torch.randn(ys.shape, generator=g)
    ys.backward(dy)
    d2 = _ffi.conv_desc(n=n, hin=h, win=w, cin=cout, hout=h, wout=w, cout=cin, ks=2, stride=1, pad=0, nsub=4, out_h=2 * h, out_w=2 * w,
                        out_sy=2, out_sx=2, out_sub=1)
    dx = ops.empty_nhwc(n, cin, 2 * h, 2 * w, DEV)
    ops.conv_forward(d2, dev(dy), _pack_phases(ws, cin, cout, (9, cin * 9, 3, 1), 3), y=dx)
    close(dx, xs.grad, what="phase data gradient of conv3x3 s2")


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 4, 16, 32, 32), (16, 1, 16, 64, 64), (3, 4, 32, 20, 12), (2, 1, 16, 9, 7), (16, 4, 16, 128, 128)])
def test_conv3x3_small_cin_k_packed_taps(n, cin, cout, h, w):
    """CTL_IN_C4: first-layer convs (1 or 4 input channels) with the 3x3 taps packed into the MFMA k dimension (pack mode 4)."""
    g = torch.Generator().manual_seed(cin * 31 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    total = ((cout + 15) // 16) * 3 * 256
    table = torch.tensor([[0, 0, cout, cin, 3, 0, cin * 9, 9, 3, 1, total, 4]], dtype=torch.int64, device=DEV)
    wp = torch.zeros(total, device=DEV)
    wd = dev(wt).contiguous()
    check(lib.ctl_pack_weights_batched(wd.data_ptr(), wp.data_ptr(), table.data_ptr(), 1, total, ops.stream_ptr()))
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, in_mode=_ffi.IN_C4, epi_flags=_ffi.EPI_BIAS | _ffi.EPI_STATS)
    xin = dev(x) if cin > 1 else x.to(DEV).contiguous()
    y, st = ops.conv_forward(d, xin, wp, bias=dev(b), want_stats=True)
    ref = F.conv2d(x, wt, b, padding=1)
    close(y, ref, what="K-packed first-layer conv")
    part = st.cpu().double().view(-1, 2, cout).sum(0)
    assert float((part[0] - ref.double().sum((0, 2, 3))).abs().max()) <= 2e-4 * float(ref.double().sum((0, 2, 3)).abs().max()) + 1e-2
    # weight gradient with (tap, channel) pairs packed into the MFMA rows
    xg = x.clone().requires_grad_(True)
    wg = wt.clone().requires_grad_(True)
    dy = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(xg, wg, padding=1).backward(dy)
    dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
    dd = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, in_mode=_ffi.IN_C4)
    ops.conv_wgrad(dd, xin, dev(dy), dw, (cin * 9, 9, 3, 1), dbias=db)
    close(dw, wg.grad, what="row-packed first-layer wgrad")
    close(db, dy.sum((0, 2, 3)), what="first-layer bias grad")


@pytest.mark.parametrize("n,c,cout,h,w,groups", [(2, 32, 16, 24, 20, 1), (16, 16, 16, 64, 64, 1), (4, 64, 32, 40, 36, 2), (32, 16, 16, 64, 64, 2)])
def test_conv_epilogue_bn_backward_reduction(n, c, cout, h, w, groups):
    """CTL_EPI_BNBWD: y = conv(x) * leaky'(u*scale+shift) and the partials hold (sum y, sum y*u) per BatchNorm group."""
    g = torch.Generator().manual_seed(c + h + groups)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(cout, c, 3, 3, generator=g) * 0.2
    u = torch.randn(n, cout, h, w, generator=g)
    scale, shift = torch.rand(groups, cout, generator=g) + 0.5, torch.randn(groups, cout, generator=g) * 0.3
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=cout, ks=3, groups=groups,
                       epi_flags=_ffi.EPI_BNBWD | _ffi.EPI_STATS, epi_slope=0.2)
    y, st = ops.conv_forward(d, dev(x), ops.pack_oihw_fwd(dev(wt)), res=dev(u), res_scale=dev(scale), res_shift=dev(shift),
                             want_stats=True)
    gi = torch.arange(n) // (n // groups)
    sa = u * scale[gi].view(n, cout, 1, 1) + shift[gi].view(n, cout, 1, 1)
    ref = F.conv2d(x, wt, padding=1) * torch.where(sa > 0, 1.0, 0.2)
    close(y, ref, what="conv+bnbwd g")
    blocks = lib.ctl_conv_stats_blocks(_ffi.desc_ptr(d))
    part = st.cpu().double().view(groups, blocks, 2, cout).sum(1)
    for k in range(groups):
        sel = gi == k
        r0, r1 = ref[sel].double().sum((0, 2, 3)), (ref[sel].double() * u[sel].double()).sum((0, 2, 3))
        assert float((part[k, 0] - r0).abs().max()) <= 2e-4 * float(r0.abs().max()) + 1e-2, "sum g"
        assert float((part[k, 1] - r1).abs().max()) <= 2e-4 * float(r1.abs().max()) + 1e-2, "sum g*u"


@pytest.mark.parametrize("n,c,cout,h,w", [(2, 16, 16, 32, 32), (2, 32, 32, 17, 23), (4, 64, 64, 16, 16), (2, 128, 128, 6, 6),
                                           (16, 16, 16, 128, 128), (2, 32, 64, 80, 72), (2, 64, 32, 50, 90)])
def test_conv3x3_s2_forward_and_zero_insert_dgrad(n, c, cout, h, w):
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, c, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=ho, wout=wo, cout=cout, ks=3, stride=2, epi_flags=_ffi.EPI_BIAS)
    y, _ = ops.conv_forward(d, dev(x.detach()), ops.pack_oihw_fwd(dev(wt)), bias=dev(b))
    ref = F.conv2d(x, wt, b, stride=2, padding=1)
    close(y, ref, what="conv3x3 s2")
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    dd = _ffi.conv_desc(n=n, hin=ho, win=wo, cin=cout, hout=h, wout=w, cout=c, ks=3, in_mode=_ffi.IN_ZINS2)
    dx, _ = ops.conv_forward(dd, dev(dy), ops.pack_oihw_dgrad(dev(wt)))
    close(dx, x.grad, what="conv3x3 s2 dgrad (zero-insert)")


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 16, 16, 16, 16), (2, 128, 64, 4, 4), (3, 32, 16, 24, 24), (16, 16, 16, 64, 64),
                                              (2, 64, 32, 40, 36), (1, 32, 32, 21, 50)])
def test_conv3x3_on_nearest_upsampled_input_and_dgrad(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=2 * h, wout=2 * w, cout=cout, ks=3, in_mode=_ffi.IN_UP2)
    y, _ = ops.conv_forward(d, dev(x.detach()), ops.pack_oihw_fwd(dev(wt)))
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, padding=1)
    close(y, ref, what="conv3x3 up2")
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    # dgrad at full resolution, then 2x2 sum-pool
    dd = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=cout, hout=2 * h, wout=2 * w, cout=cin, ks=3)
    dup, _ = ops.conv_forward(dd, dev(dy), ops.pack_oihw_dgrad(dev(wt)))
    dx = torch.empty((n, cin, h, w), device=DEV).contiguous(memory_format=torch.channels_last)
    check(lib.ctl_sumpool2(dup.data_ptr(), dx.data_ptr(), n, h, w, cin, 0, ops.stream_ptr()))
    close(dx, x.grad, what="up2 dgrad")


@pytest.mark.parametrize("n,cin,cout,h,w,up", [(2, 16, 32, 32, 32, 0), (2, 128, 128, 4, 4, 0), (2, 64, 32, 8, 8, 1),
                                                (2, 16, 4, 64, 64, 0), (2, 16, 1, 32, 32, 0), (16, 16, 16, 64, 64, 1),
                                                (2, 64, 32, 48, 80, 0), (2, 32, 32, 24, 40, 1)])
def test_conv1x1_residual_epilogue(n, cin, cout, h, w, up):
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    v = torch.randn(n, cout, ho, wo, generator=g)
    rs, rh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=ho, wout=wo, cout=cout, ks=1, in_mode=_ffi.IN_UP2 if up else 0,
                       epi_flags=_ffi.EPI_BIAS | _ffi.EPI_RES, epi_act=_ffi.ACT_LEAKY, epi_slope=0.2)
    y, _ = ops.conv_forward(d, dev(x), ops.pack_oihw_fwd(dev(wt)), bias=dev(b), res=dev(v), res_scale=dev(rs), res_shift=dev(rh))
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = leaky(F.conv2d(xi, wt, b) + v * rs.view(1, -1, 1, 1) + rh.view(1, -1, 1, 1), 0.2)
    close(y, ref, what="conv1x1+res+leaky")
    # sigmoid epilogue and accumulate epilogue
    d2 = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=ho, wout=wo, cout=cout, ks=1, in_mode=_ffi.IN_UP2 if up else 0,
                        epi_flags=_ffi.EPI_BIAS, epi_act=_ffi.ACT_SIGMOID)
    y2, _ = ops.conv_forward(d2, dev(x), ops.pack_oihw_fwd(dev(wt)), bias=dev(b))
    close(y2, torch.sigmoid(F.conv2d(xi, wt, b)), what="conv1x1+sigmoid")
    d3 = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=ho, wout=wo, cout=cout, ks=1, in_mode=_ffi.IN_UP2 if up else 0,
                        epi_flags=_ffi.EPI_ACCUM)
    y3 = dev(v.clone())
    ops.conv_forward(d3, dev(x), ops.pack_oihw_fwd(dev(wt)), y=y3)
    close(y3, F.conv2d(xi, wt) + v, what="conv1x1 accumulate")


@pytest.mark.parametrize("n,c,h,w", [(2, 16, 16, 16), (2, 128, 4, 4), (3, 32, 9, 7), (16, 16, 64, 64), (2, 32, 40, 56)])
def test_conv_transpose2x2_forward_dgrad_wgrad(n, c, h, w):
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(c, c, 2, 2, generator=g) * 0.2).requires_grad_(True)   # [Cin][Cout][2][2]
    b = torch.randn(c, generator=g, requires_grad=True)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    wdev = wt.detach().to(DEV).contiguous()          # raw [Cin][Cout][2][2] memory (not channels_last)
    sub = lib.ctl_conv_wpack_floats(c, c, 1)
    wp = torch.empty(4 * sub, device=DEV)
    for z in range(4):
        check(lib.ctl_pack_weights(wdev.data_ptr() + 4 * z, wp.data_ptr() + 4 * z * sub, c, c, 1, 4, c * 4, 0, 0, 0, ops.stream_ptr()))
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=c, ks=1, epi_flags=_ffi.EPI_BIAS, out_h=2 * h, out_w=2 * w,
                       out_sy=2, out_sx=2, nsub=4, out_sub=1)
    y, _ = ops.conv_forward(d, dev(x.detach()), wp, bias=dev(b.detach()))
    close(y, ref, what="convT fwd")
    # dgrad: conv 2x2 stride 2 pad 0 over dy with W_eff[co'=ci][ci'=co][a,b] = Wt[ci][co][a][b]
    wpd = ops.pack_weights(wdev, c, c, 2, (c * 4, 4, 2, 1), False)
    dd = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=c, hout=h, wout=w, cout=c, ks=2, stride=2, pad=0)
    dx, _ = ops.conv_forward(dd, dev(dy), wpd)
    close(dx, x.grad, what="convT dgrad")
    # wgrad by role swap: "input" = dy (full res), "output grad" = x
    dw = torch.zeros(c, c, 2, 2, device=DEV)
    ops.conv_wgrad(dd, dev(dy), dev(x.detach()), dw, (c * 4, 4, 2, 1))
    close(dw, wt.grad, what="convT wgrad")
    # bias grad: channel sums
    part = torch.empty(_ffi.RED_BLOCKS * 2 * c, device=DEV)
    db = torch.zeros(c, device=DEV)
    dyd = dev(dy)
    check(lib.ctl_bwd_reduce(2, dyd.data_ptr(), None, None, None, None, 0.0, n * 4 * h * w, c, part.data_ptr(), 1, ops.stream_ptr()))
    check(lib.ctl_chan_sum_finalize(part.data_ptr(), c, db.data_ptr(), 0, ops.stream_ptr()))
    close(db, b.grad, what="convT bias grad")


WGRAD_CASES = [
    # n, cin, cout, h, w, ks, stride, up
    (2, 16, 16, 32, 32, 3, 1, 0), (16, 16, 16, 64, 64, 3, 1, 0), (2, 32, 64, 16, 16, 3, 1, 0), (2, 128, 128, 8, 8, 3, 1, 0),
    (2, 1, 16, 32, 32, 3, 1, 0), (2, 4, 16, 20, 12, 3, 1, 0), (2, 64, 64, 17, 23, 3, 2, 0), (4, 16, 16, 64, 64, 3, 2, 0),
    (2, 64, 32, 8, 8, 3, 1, 1), (2, 16, 32, 32, 32, 1, 1, 0), (2, 128, 64, 4, 4, 1, 1, 1), (2, 16, 4, 64, 64, 1, 1, 0),
    (2, 16, 1, 32, 32, 1, 1, 0), (4, 16, 16, 128, 128, 3, 1, 0),
    # interior tiles of the staging fast path, several chunks
    (2, 32, 32, 48, 40, 3, 1, 0), (2, 64, 32, 20, 28, 3, 1, 1), (2, 32, 48, 50, 70, 3, 2, 0), (2, 32, 32, 24, 40, 1, 1, 1),
]


@pytest.mark.parametrize("n,cin,cout,h,w,ks,stride,up", WGRAD_CASES)
def test_conv_wgrad(n, cin, cout, h, w, ks, stride, up):
    g = torch.Generator().manual_seed(cin * 7 + cout + h + ks)
    x = torch.randn(n, cin, h, w, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = (torch.randn(cout, cin, ks, ks, generator=g) * 0.2).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    xin = leaky(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wt, b, stride=stride, padding=1 if ks == 3 else 0)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    ho, wo = ref.shape[2:]
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=ho, wout=wo, cout=cout, ks=ks, stride=stride,
                       in_mode=_ffi.IN_UP2 if up else 0, pro_affine=1, pro_slope=0.2)
    dw = torch.full((cout, cin, ks, ks), 7.0, device=DEV)
    db = torch.full((cout,), 7.0, device=DEV)
    ops.conv_wgrad(d, dev(x), dev(dy), dw, (cin * ks * ks, ks * ks, ks, 1), dbias=db, pro_scale=dev(sc), pro_shift=dev(sh))
    close(dw, wt.grad, rel=3e-4, what="wgrad")
    close(db, b.grad, rel=3e-4, what="bias grad")
    ops.conv_wgrad(d, dev(x), dev(dy), dw, (cin * ks * ks, ks * ks, ks, 1), dbias=db, pro_scale=dev(sc), pro_shift=dev(sh),
                   accumulate=True)
    close(dw, 2 * wt.grad, rel=3e-4, what="wgrad accumulate")


@pytest.mark.parametrize("n,c,h,w", [(2, 16, 32, 32), (16, 16, 256, 256), (2, 128, 4, 4), (3, 32, 9, 7)])
def test_batchnorm_forward_stats_and_backward(n, c, h, w):
    g = torch.Generator().manual_seed(c + h)
    u = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.7).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(c, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    a = F.leaky_relu(F.batch_norm(u, rm, rv, gamma, beta, True, 0.1, 1e-5), 0.2)
    da = torch.randn(a.shape, generator=g)
    a.backward(da)
    # forward: stats via the identity 1x1 conv path is covered elsewhere; here feed exact partials from a reduce kernel
    ud = dev(u.detach())
    M = n * h * w
    part = torch.empty(lib.ctl_bwd_reduce_rows(0, M, c) * 2 * c, device=DEV)      # exactly the rows the reduction writes (bn_finalize reads them all)
    # sum u and sum u*u through bwd_reduce mode 0 with act_src=ones (leaky'(1)=1): g=u, sums: sum u, sum u*u
    ones = torch.ones_like(ud)
    check(lib.ctl_bwd_reduce(0, ud.data_ptr(), ones.data_ptr(), ud.data_ptr(), None, None, 0.2, M, c, part.data_ptr(), 1, ops.stream_ptr()))
    rmd, rvd, nbt = dev(rm0), dev(rv0), torch.zeros(1, dtype=torch.int64, device=DEV)
    scale, shift, mean, invstd = ops.bn_finalize(part, c, M, dev(gamma.detach()), dev(beta.detach()), running_mean=rmd,
                                                 running_var=rvd, nbt=nbt)
    close(mean, u.detach().mean((0, 2, 3)), rel=1e-5, what="mean")
    close(invstd, 1 / torch.sqrt(u.detach().var((0, 2, 3), unbiased=False) + 1e-5), rel=1e-5, what="invstd")
    close(rmd, rm, rel=1e-5, what="running_mean")
    close(rvd, rv, rel=1e-5, what="running_var")
    assert int(nbt.item()) == 1
    close(ops.bn_act(ud, scale, shift, 0.2), a, rel=1e-5, what="bn_act")
    # backward: reduce -> finalize -> apply (mode 1)
    dad = dev(da)
    coef = torch.empty(3 * c, device=DEV)
    dgamma, dbeta = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    check(lib.ctl_bwd_reduce(1, dad.data_ptr(), None, ud.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0.2, M, c, part.data_ptr(), 1, ops.stream_ptr()))
    gd = dev(gamma.detach())
    check(lib.ctl_bn_bwd_finalize(part.data_ptr(), c, M, gd.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                  coef.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0, 1, 0, ops.stream_ptr()))
    du = torch.empty_like(ud)
    check(lib.ctl_bwd_apply(1, dad.data_ptr(), None, ud.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0.2, coef.data_ptr(), M, c,
                            None, du.data_ptr(), 1, ops.stream_ptr()))
    close(du, u.grad, rel=5e-4, what="BN backward dx")
    close(dgamma, gamma.grad, rel=5e-4, what="dgamma")
    close(dbeta, beta.grad, rel=5e-4, what="dbeta")


def test_residual_tail_backward():
    g = torch.Generator().manual_seed(3)
    n, c, h, w = 2, 32, 16, 16
    v = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    r = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    out = F.leaky_relu(r + F.batch_norm(v, None, None, gamma, beta, True, 0.1, 1e-5), 0.2)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    M = n * h * w
    vd, outd, doutd = dev(v.detach()), dev(out.detach()), dev(dout)
    mean = v.detach().mean((0, 2, 3))
    invstd = 1 / torch.sqrt(v.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    part = torch.empty(_ffi.RED_BLOCKS * 2 * c, device=DEV)
    coef = torch.empty(3 * c, device=DEV)
    gd, md, isd = dev(gamma), dev(mean), dev(invstd)     # keep alive: raw pointers below
    check(lib.ctl_bwd_reduce(0, doutd.data_ptr(), outd.data_ptr(), vd.data_ptr(), None, None, 0.2, M, c, part.data_ptr(), 1, ops.stream_ptr()))
    check(lib.ctl_bn_bwd_finalize(part.data_ptr(), c, M, gd.data_ptr(), md.data_ptr(), isd.data_ptr(),
                                  coef.data_ptr(), None, None, 0, 1, 0, ops.stream_ptr()))
    ds, dv = torch.empty_like(vd), torch.empty_like(vd)
    check(lib.ctl_bwd_apply(0, doutd.data_ptr(), outd.data_ptr(), vd.data_ptr(), None, None, 0.2, coef.data_ptr(), M, c,
                            ds.data_ptr(), dv.data_ptr(), 1, ops.stream_ptr()))
    close(ds, r.grad, rel=1e-5, what="ds")
    close(dv, v.grad, rel=5e-4, what="dv")


@pytest.mark.parametrize("c", [4, 3, 7])          # 4: the 16-byte-row kernels of the 4-class maps; others: runtime channel count
def test_stn_input_losses_argmax(c):
    g = torch.Generator().manual_seed(4)
    n, h, w = 3, 24, 20
    x = (torch.randn(n, c, h, w, generator=g) * 3).requires_grad_(True)
    lab = torch.randint(0, c, (n, h, w), generator=g)
    p = torch.softmax(x / 2, dim=1)
    dp = torch.randn(p.shape, generator=g)
    p.backward(dp)
    pd = ops.softmax_t_fwd(dev(x.detach()), 2.0)
    close(pd, p, rel=1e-6, what="softmax T=2")
    close(ops.softmax_t_bwd(pd, dev(dp), 2.0), x.grad, rel=1e-5, what="softmax bwd")
    assert torch.equal(ops.onehot(dev(lab), c).cpu(), F.one_hot(lab, c).permute(0, 3, 1, 2).float())
    x.grad = None
    loss = F.cross_entropy(x, lab)
    (loss * 0.7).backward()
    ld = ops.ce2d_fwd(dev(x.detach()), dev(lab))
    assert abs(float(ld) - float(loss)) < 2e-6
    close(ops.ce2d_bwd(dev(x.detach()), dev(lab), torch.tensor(0.7, device=DEV)), x.grad, rel=1e-5, what="ce bwd")
    a = torch.rand(n, 1, h, w, generator=g, requires_grad=True)
    b = torch.rand(n, 1, h, w, generator=g)
    l2 = 0.5 * F.mse_loss(a, b)
    l2.backward()
    assert abs(float(ops.mse_fwd(dev(a.detach()), dev(b), 0.5)) - float(l2)) < 1e-7
    close(ops.mse_bwd(dev(a.detach()), dev(b), torch.tensor(1.0, device=DEV), 0.5), a.grad, rel=1e-5, what="mse bwd")
    assert torch.equal(ops.argmax_c(dev(x.detach())).cpu(), x.detach().max(1)[1].to(torch.uint8))
    y = torch.sigmoid(a.detach())
    dyy = torch.randn(y.shape, generator=g)
    yd, dyd = dev(y), dev(dyy)
    dx = torch.empty_like(yd)
    check(lib.ctl_sigmoid_bwd(dyd.data_ptr(), yd.data_ptr(), dx.data_ptr(), y.numel(), ops.stream_ptr()))
    close(dx, dyy * y * (1 - y), rel=1e-6, what="sigmoid bwd")


def test_adam_matches_torch():
    g = torch.Generator().manual_seed(5)
    p = torch.randn(10007, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4)
    pd, m, v = p.to(DEV), torch.zeros(10007, device=DEV), torch.zeros(10007, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g) * 0.1
        ref.grad = gr.clone()
        opt.step()
        ops.adam_step(pd, gr.to(DEV), m, v, 1e-4, 0.9, 0.999, 1e-8, step)
        assert float((pd.cpu() - ref.detach()).abs().max()) < 2e-7


@pytest.mark.parametrize("n,c,h,w", [(16, 128, 16, 16), (2, 128, 4, 4), (3, 64, 6, 5), (4, 256, 32, 32), (3, 32, 48, 40)])
def test_latent_mask_kernels(n, c, h, w):
    g = torch.Generator().manual_seed(c + h)
    grad = torch.randn(n, c, h, w, generator=g)
    code = torch.rand(n, c, h, w, generator=g)
    for mode in (0, 1):
        L = c if mode == 0 else h * w
        ref_score = grad.view(n, c, -1).mean(2) if mode == 0 else grad.mean(1).reshape(n, h * w)
        score = ops.latent_score(dev(grad), mode)
        close(score, ref_score, rel=2e-6, what="score")
        s_cpu = score.cpu()
        for k, soft in ((0, False), (L // 2, False), (L - 1, True), (int(L * 0.37), True)):
            noise = torch.rand(n, L, generator=g) if soft else None
            masked, mask = ops.latent_mask_apply(dev(code), score, mode, k, None if noise is None else dev(noise))
            # select logic is integer-exact given the same scores (model_util.py:231-244)
            thr = torch.sort(s_cpu, dim=1, descending=True)[0][:, k].view(-1, 1)
            hit = s_cpu > thr
            vec = torch.where(hit, 0.5 * noise if soft else torch.zeros_like(s_cpu), torch.ones_like(s_cpu))
            ref_mask = vec.view(n, c, 1, 1) if mode == 0 else vec.view(n, 1, h, w)
            assert torch.equal(mask.cpu(), ref_mask), (mode, k, soft)
            assert torch.equal(masked.cpu(), code * ref_mask)
            assert int(hit.sum(1).max()) <= k
        # k through a device scalar (graph-replay form)
        kd = torch.tensor([L // 3], dtype=torch.int32, device=DEV)
        m1 = ops.latent_mask_apply(dev(code), score, mode, kd)[1]
        m2 = ops.latent_mask_apply(dev(code), score, mode, L // 3)[1]
        assert torch.equal(m1, m2)
    # ties: equal scores must all stay unmasked under the strict '>' (e.g. dead channels with zero gradient)
    sc = torch.zeros(2, 8)
    sc[:, 0] = 1.0
    cd = torch.ones(2, 8, 2, 2)
    masked, mask = ops.latent_mask_apply(dev(cd), dev(sc), 0, 3)
    assert mask.cpu().view(2, 8).tolist() == [[0.0] + [1.0] * 7] * 2


@pytest.mark.parametrize("n,c,h,w", [(16, 128, 16, 16), (2, 128, 4, 4), (3, 64, 6, 5), (4, 256, 32, 32), (3, 32, 48, 40), (2, 16, 64, 64),
                                     (1, 8, 3, 3), (40, 128, 64, 64), (1200, 8, 4, 4)])
def test_latent_mask_fused_is_bitwise_the_three_launch_path(n, c, h, w):
    """ONE launch (score -> grid barrier -> rank-select -> apply) == ctl_latent_score + ctl_latent_mask_apply: scores, masks and masked
    codes bit for bit, for both modes, hard / soft masks, host / device k, ragged sizes and grids that span several waves of blocks;
    repeated launches reuse the barrier's sync words."""
    g = torch.Generator().manual_seed(c + h)
    grad, code = dev(torch.randn(n, c, h, w, generator=g)), dev(torch.rand(n, c, h, w, generator=g))
    for mode in (0, 1):
        L = c if mode == 0 else h * w
        if L > 1024 and mode == 1:
            ks = [L // 3]                               # (long rows: the two-call path takes its threshold from a bitonic sort)
        else:
            ks = [0, L // 2, L - 1, int(L * 0.37)]
        score = ops.latent_score(grad, mode)
        for rep, k in enumerate(ks):
            noise = dev(torch.rand(n, L, generator=g)) if rep % 2 else None
            m_ref, k_ref = ops.latent_mask_apply(code, score, mode, k, noise)
            kk = torch.tensor([k], dtype=torch.int32, device=DEV) if rep == 1 else k
            m, km, sc = ops.latent_mask(grad, code, mode, kk, noise, want_score=True)
            assert torch.equal(sc, score), (mode, k)
            assert torch.equal(km, k_ref) and torch.equal(m, m_ref), (mode, k)
    with pytest.raises(_ffi.CtlError):
        ops.latent_mask(grad, code, 0, c)               # k out of range


def test_dropout2d_injected_and_device_rng():
    g = torch.Generator().manual_seed(6)
    z = torch.rand(4, 128, 8, 8, generator=g)
    keep = (torch.rand(4, 128, generator=g) > 0.5).float()
    out, ko = ops.dropout2d(dev(z), 0.5, keep=dev(keep))
    assert torch.equal(out.cpu(), z * keep.view(4, 128, 1, 1) * 2.0) and torch.equal(ko.cpu(), keep)
    out2, k2 = ops.dropout2d(dev(z), 0.5, seed=1234)
    out3, k3 = ops.dropout2d(dev(z), 0.5, seed=1234)
    assert torch.equal(k2, k3) and torch.equal(out2, out3)
    frac = float(k2.mean())
    assert 0.35 < frac < 0.65 and set(k2.cpu().unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(out2.cpu(), z * k2.cpu().view(4, 128, 1, 1) * 2.0)
    u = ops.uniform((1 << 16,), DEV, 99).cpu()
    assert 0 <= float(u.min()) and float(u.max()) < 1 and abs(float(u.mean()) - 0.5) < 0.01


# ---------------------------------------------------------------------------------------------- SURVEY 8(f) rows 1 and 3
def _io_cases():
    import os
    return torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_cases.pt"), weights_only=False)


def test_input_pipeline_kernels_bit_exact_vs_reference_vectors():
    io = _io_cases()
    for r in io["rescale"]:
        y = ops.rescale_intensity(r["x"].to(DEV), r["new_min"], r["new_max"])
        assert torch.equal(y.cpu(), r["y"]), float((y.cpu() - r["y"]).abs().max())
    for r in io["crop_or_pad"]:
        a, b = ops.crop_or_pad(r["image"].to(DEV), r["size"], r["label"].to(DEV))
        assert torch.equal(a.cpu(), r["image_out"]) and torch.equal(b.cpu(), r["label_out"])
        u8, _ = ops.crop_or_pad(r["label"].to(torch.uint8).to(DEV), r["size"])
        assert torch.equal(u8.cpu(), r["label_out"].to(torch.uint8))
    r = io["noise_clamp"][0]
    assert torch.equal(ops.noise_clamp(r["clean"].to(DEV), r["noise"].to(DEV)).cpu(), r["out"])
    # device RNG: reproducible from the seed, ~N(0, sigma^2) before clamping, bounds respected
    x = torch.full((64, 1, 128, 128), 0.5, device=DEV)
    a, b, c = ops.noise_clamp(x, sigma=0.05, seed=3), ops.noise_clamp(x, sigma=0.05, seed=3), ops.noise_clamp(x, sigma=0.05, seed=4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    dlt = (a - 0.5).double()
    assert abs(float(dlt.mean())) < 3e-4 and abs(float(dlt.std()) - 0.05) < 5e-4
    wide = ops.noise_clamp(x, sigma=2.0, seed=5)
    assert float(wide.min()) == 0.0 and float(wide.max()) == 1.0


def test_confusion_matrix_on_device_vs_reference_vectors():
    from cooperative_training_and_latent_space_data_augmentation_amd.metrics import runningScore, dice_from_confusion
    r = _io_cases()["running_score"][0]
    rs = runningScore(4)
    for lt, lp in r["batches"]:
        rs.update(lt.to(DEV), lp.to(DEV))
    assert torch.equal(rs._dev_hist.cpu(), r["confusion"].long())          # integer-exact
    score, iu = rs.get_scores()
    for k, v in r["score"].items():
        assert score[k] == v or (np.isnan(score[k]) and np.isnan(v))
    for k, v in r["cls_iu"].items():
        assert iu[k] == v or (np.isnan(iu[k]) and np.isnan(v))
    # a big one: 16 x 256 x 256 labels, against numpy bincount
    g = torch.Generator().manual_seed(0)
    lt = torch.randint(-1, 5, (16, 256, 256), generator=g)
    lp = torch.randint(0, 4, (16, 256, 256), generator=g).to(torch.uint8)
    h = ops.confusion_hist(lt.to(DEV), lp.to(DEV), 4).cpu().numpy()
    keep = (lt >= 0) & (lt < 4)
    ref = np.bincount((4 * lt[keep] + lp[keep].long()).numpy(), minlength=16).reshape(4, 4)
    assert np.array_equal(h, ref)
    d = dice_from_confusion(h)
    assert all(0.0 <= v <= 1.0 for v in d)


def test_basic_operations_module_matches_upstream_return_convention():
    from cooperative_training_and_latent_space_data_augmentation_amd import basic_operations as B
    io = _io_cases()
    r = io["crop_or_pad"][0]                      # (2, 11, 14) -> (8, 8): crop both axes
    out = B.crop_or_pad(r["image"].to(DEV), r["size"], r["label"].to(DEV))
    assert len(out) == 6 and torch.equal(out[0].cpu(), r["image_out"]) and out[2:] == ((11 - 8) // 2, (14 - 8) // 2, 11, 14)
    r = io["crop_or_pad"][3]                      # already the right size: upstream returns the pair unchanged
    out = B.crop_or_pad(r["image"].to(DEV), r["size"], r["label"].to(DEV))
    assert len(out) == 2 and torch.equal(out[0].cpu(), r["image_out"])
    r = io["rescale"][0]
    assert torch.equal(B.rescale_intensity(r["x"].to(DEV)).cpu(), r["y"])
    n = io["noise_clamp"][0]
    assert torch.equal(B.add_input_noise(n["clean"].to(DEV), noise=n["noise"].to(DEV)).cpu(), n["out"])


def test_patient_wise_scores_device_path_vs_reference():
    """runningMySegmentationScore with device tensors (voxel counts from the confusion-matrix kernel) == the reference's rows."""
    from cooperative_training_and_latent_space_data_augmentation_amd.metrics import runningMySegmentationScore
    for r in _io_cases()["patient_scores"]:
        ms = runningMySegmentationScore(4, idx2cls_dict=None if r["foreground_only"] else r["idx2cls"],
                                        metrics_list=["Dice", "VolError", "VolSim"], foreground_only=r["foreground_only"])
        for k, ((pr, gt), row) in enumerate(zip(r["volumes"], r["rows"])):
            assert ms.update("p%d" % k, pr.cuda(), gt.cuda()) == row
        assert ms.get_scores()[0] == r["summary"]
    # a ground-truth label outside [0, n): belongs to no class, the prediction under it still counts (metrics.py:205-223)
    rng = np.random.RandomState(3)
    gt = rng.randint(-1, 6, (4, 16, 16)).astype(np.int64)
    pr = rng.randint(0, 4, gt.shape).astype(np.uint8)
    ms = runningMySegmentationScore(4, metrics_list=["Dice", "VolError"])
    from oracle import ref_cpu as O_
    assert ms.update("x", torch.from_numpy(pr).cuda(), torch.from_numpy(gt).cuda())[1:] == O_.patient_scores(pr, gt, range(4), ("Dice", "VolError"))
    with pytest.raises(ValueError):
        runningMySegmentationScore(4, foreground_only=True).update("x", torch.from_numpy(pr).cuda(), torch.from_numpy(gt).cuda())


def test_surface_scores_from_device_volumes():
    """'HD' / 'ASD' with device volumes: counts from the confusion kernel, surface distances on the host copy -- same rows."""
    from cooperative_training_and_latent_space_data_augmentation_amd.metrics import runningMySegmentationScore
    for r in _io_cases()["surface_scores"]:
        ms = runningMySegmentationScore(4, idx2cls_dict=None if r["foreground_only"] else r["idx2cls"],
                                        metrics_list=["Dice", "HD", "ASD"], foreground_only=r["foreground_only"])
        for k, ((pr, gt), row) in enumerate(zip(r["volumes"], r["rows"])):
            got = ms.update("s%d" % k, pr.cuda(), gt.cuda(), voxel_spacing=r["spacing"])
            assert np.allclose(got[1:], row[1:], rtol=0, atol=1e-12)
