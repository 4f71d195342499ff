"""bf16 kernel family (BASELINE config 3: bf16 activation storage + v_mfma_f32_16x16x32_bf16, fp32 accumulate / statistics / master
weights) against PyTorch with THE SAME ROUNDING POINTS: MFMA operands (activations after the fp32 BatchNorm + LeakyReLU prologue,
weights) rounded to bf16 (RNE), accumulation in fp32 (the reference accumulates the rounded operands in fp64), stored outputs rounded to
bf16 where the tensor is stored as bf16.  Tolerances: fp32 outputs 3e-4 of max|ref| (fp32 accumulation order over K <= 1152 products);
bf16-stored outputs additionally one bf16 rounding of the result: 2^-8 relative per element; with the fused prologue 1e-3 (the kernel's
fma and the reference's mul + add differ by an fp32 ulp, which now and then flips the bf16 rounding of one operand element)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check  # noqa: E402

DEV = "cuda"
BF = _ffi.DT_BF16


def rb(t):
    """round to bf16 (RNE), back to fp64 for the reference arithmetic"""
    return t.float().to(torch.bfloat16).double()


def dev(x, bf16=False):
    x = x.to(DEV)
    if bf16:
        x = x.to(torch.bfloat16)
    return x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x.contiguous()


def leaky(x, s):
    return torch.where(x > 0, x, x * s)


def close(a, b, rel, what, bf16_out=False):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = rel * max(float(b.abs().max()), 1e-6) + 1e-7
    err = (a - b).abs()
    if bf16_out:
        err = err - b.abs() * 2.0 ** -8                  # one bf16 rounding of the stored result
    assert float(err.max()) <= tol, f"{what}: max err {float(err.max()):.3e} > tol {tol:.3e}"


CASES = [(2, 16, 16, 32, 32), (4, 16, 16, 64, 64), (2, 32, 64, 16, 16), (2, 128, 128, 8, 8), (2, 64, 32, 24, 20), (1, 32, 32, 6, 6),
         (2, 32, 32, 48, 48), (1, 64, 48, 40, 72), (3, 48, 16, 70, 70), (16, 16, 16, 128, 128)]


@pytest.mark.parametrize("x16,y16", [(False, False), (True, True), (False, True), (True, False)])
@pytest.mark.parametrize("n,cin,cout,h,w", CASES)
def test_bf16_conv3x3_s1(n, cin, cout, h, w, x16, y16):
    g = torch.Generator().manual_seed(n * 1000 + cin * 10 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    if x16:
        x = x.to(torch.bfloat16).float()                 # the stored tensor IS bf16
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wp = ops.pack_oihw_fwd_bf16(dev(wt))
    dt = BF | (_ffi.DT_X16 if x16 else 0) | (_ffi.DT_Y16 if y16 else 0)
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, epi_flags=_ffi.EPI_BIAS | _ffi.EPI_STATS, dt=dt)
    y, stats = ops.conv_forward(d, dev(x, x16), wp, bias=dev(b), want_stats=True)
    assert y.dtype == (torch.bfloat16 if y16 else torch.float32)
    ref = F.conv2d(rb(x), rb(wt), b.double(), padding=1)
    close(y, ref, 3e-4, "conv3x3 bf16", y16)
    st = stats.view(-1, 2, cout).double().sum(0).cpu()   # statistics: from the fp32 accumulators, before the output rounding
    close(st[0], ref.sum((0, 2, 3)), 2e-4, "stats sum")
    close(st[1], (ref ** 2).sum((0, 2, 3)), 2e-4, "stats sumsq")
    d2 = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, epi_flags=_ffi.EPI_BIAS, pro_affine=1, pro_slope=0.2, dt=dt)
    y2, _ = ops.conv_forward(d2, dev(x, x16), wp, bias=dev(b), pro_scale=dev(sc), pro_shift=dev(sh))
    pro = leaky(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2)      # fp32 prologue, THEN the operand rounding
    close(y2, F.conv2d(rb(pro), rb(wt), b.double(), padding=1), 1e-3, "conv3x3 bf16 + prologue", y16)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 1, 16, 32, 32), (2, 4, 16, 20, 12), (16, 1, 16, 64, 64), (2, 16, 4, 32, 32), (3, 16, 1, 16, 16),
                                             (2, 128, 64, 3, 3)])
def test_bf16_conv3x3_network_boundaries(n, cin, cout, h, w):
    """fp32 network inputs with 1 / 4 channels (first layers) and fp32 outputs with 1 / 4 channels (last layers)."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    y16 = cout % 4 == 0 and cout >= 16
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, epi_flags=_ffi.EPI_BIAS | _ffi.EPI_STATS,
                       dt=BF | (_ffi.DT_Y16 if y16 else 0))
    y, stats = ops.conv_forward(d, dev(x), ops.pack_oihw_fwd_bf16(dev(wt)), bias=dev(b), want_stats=True)
    ref = F.conv2d(rb(x), rb(wt), b.double(), padding=1)
    close(y, ref, 3e-4, "boundary conv", y16)
    close(stats.view(-1, 2, cout).double().sum(0).cpu()[0], ref.sum((0, 2, 3)), 2e-4, "stats sum")


@pytest.mark.parametrize("n,c,cout,h,w", [(2, 16, 16, 32, 32), (2, 32, 64, 16, 24), (4, 64, 128, 8, 8), (16, 16, 32, 64, 64)])
def test_bf16_residual_tail_1x1_up2_and_strided(n, c, cout, h, w):
    g = torch.Generator().manual_seed(c + cout + h)
    x = torch.randn(n, c, h, w, generator=g).to(torch.bfloat16).float()
    w1 = torch.randn(cout, c, 1, 1, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    v = torch.randn(n, cout, 2 * h, 2 * w, generator=g).to(torch.bfloat16).float()
    rs, rh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    dt = BF | _ffi.DT_X16 | _ffi.DT_Y16 | _ffi.DT_RES16
    # out = LReLU(conv1x1(up2(x)) + BN(v)): the residual tail of res_up_family (nearest-upsample input mode, bf16 residual operand)
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=2 * h, wout=2 * w, cout=cout, ks=1, in_mode=_ffi.IN_UP2,
                       epi_flags=_ffi.EPI_BIAS | _ffi.EPI_RES, epi_act=_ffi.ACT_LEAKY, epi_slope=0.2, dt=dt)
    y, _ = ops.conv_forward(d, dev(x, True), ops.pack_oihw_fwd_bf16(dev(w1)), bias=dev(b), res=dev(v, True), res_scale=dev(rs), res_shift=dev(rh))
    up = F.interpolate(rb(x), scale_factor=2, mode="nearest")
    ref = leaky(F.conv2d(up, rb(w1), b.double()) + v.double() * rs.double().view(1, -1, 1, 1) + rh.double().view(1, -1, 1, 1), 0.2)
    close(y, ref, 3e-4, "1x1 + residual + leaky (up2)", True)
    # 3x3 stride 2 (res_convdown.down) and 2x2 stride 2 (ConvTranspose2d data gradient)
    w3 = torch.randn(cout, c, 3, 3, generator=g) * 0.2
    x2 = torch.randn(n, c, 2 * h, 2 * w, generator=g).to(torch.bfloat16).float()
    d3 = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=c, hout=h, wout=w, cout=cout, ks=3, stride=2, epi_flags=_ffi.EPI_BIAS, dt=BF | _ffi.DT_X16 | _ffi.DT_Y16)
    y3, _ = ops.conv_forward(d3, dev(x2, True), ops.pack_oihw_fwd_bf16(dev(w3)), bias=dev(b))
    close(y3, F.conv2d(rb(x2), rb(w3), b.double(), stride=2, padding=1), 3e-4, "3x3 stride 2", True)
    w2 = torch.randn(cout, c, 2, 2, generator=g) * 0.3
    d2 = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=c, hout=h, wout=w, cout=cout, ks=2, stride=2, pad=0, dt=BF | _ffi.DT_X16 | _ffi.DT_Y16)
    y2, _ = ops.conv_forward(d2, dev(x2, True), ops.pack_oihw_fwd_bf16(dev(w2)))
    close(y2, F.conv2d(rb(x2), rb(w2), stride=2), 3e-4, "2x2 stride 2", True)
    # data gradient of the 3x3 conv: flipped / transposed weights, accumulate into an existing bf16 tensor
    dy = torch.randn(n, cout, h, w, generator=g).to(torch.bfloat16).float()
    w33 = torch.randn(cout, c, 3, 3, generator=g) * 0.2
    acc0 = torch.randn(n, c, h, w, generator=g).to(torch.bfloat16)
    dd = _ffi.conv_desc(n=n, hin=h, win=w, cin=cout, hout=h, wout=w, cout=c, ks=3, epi_flags=_ffi.EPI_ACCUM, dt=BF | _ffi.DT_X16 | _ffi.DT_Y16)
    yacc = dev(acc0.float(), True).clone()
    ops.conv_forward(dd, dev(dy, True), ops.pack_oihw_dgrad_bf16(dev(w33)), y=yacc)
    close(yacc, F.conv_transpose2d(rb(dy), rb(w33), padding=1) + acc0.double(), 3e-4, "dgrad + accumulate", True)


def _pack_phases_bf16(w, cout_eff, cin_eff, strides, mode):
    sub = lib.ctl_conv_wpack_floats(cin_eff, cout_eff, 2)
    table = np.asarray([[0, z * sub, cout_eff, cin_eff, 2, z, *strides, sub, mode] for z in range(4)], dtype=np.int64)
    wd, td = w.to(DEV).contiguous(), torch.from_numpy(table).to(DEV)
    out = torch.zeros(4 * sub, device=DEV)
    check(lib.ctl_pack_weights_bf16_batched(wd.data_ptr(), out.data_ptr(), td.data_ptr(), 4, sub, ops.stream_ptr()))
    return out


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 16, 16, 16, 16), (2, 128, 64, 4, 4), (3, 32, 16, 24, 20), (16, 16, 16, 64, 64)])
def test_bf16_phase_convs_and_pooled_dgrad(n, cin, cout, h, w):
    """The exact re-formulations around the resampling layers (pack modes 1-3): the COMBINED weights are formed in fp32 and rounded once."""
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    dt = BF | _ffi.DT_X16 | _ffi.DT_Y16
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=2, stride=1, pad=2, nsub=4, out_h=2 * h, out_w=2 * w,
                       out_sy=2, out_sx=2, out_sub=1, epi_flags=_ffi.EPI_BIAS, dt=dt)
    y = ops.empty_nhwc(n, cout, 2 * h, 2 * w, DEV, torch.bfloat16)
    ops.conv_forward(d, dev(x, True), _pack_phases_bf16(wt, cout, cin, (cin * 9, 9, 3, 1), 2), bias=dev(b), y=y)
    # reference with the phase weights combined in fp32 and rounded once (what the pack kernel does)
    ref = torch.zeros(n, cout, 2 * h, 2 * w, dtype=torch.float64)
    xp = F.pad(rb(x), (1, 1, 1, 1))
    for a in range(2):
        for bb in range(2):
            k = torch.zeros(cout, cin, 2, 2)
            for kh in range(3):
                for kw in range(3):
                    k[:, :, (a + kh + 1) // 2 - a, (bb + kw + 1) // 2 - bb] += wt[:, :, kh, kw]      # tap of phase (a, bb) that W[kh][kw] lands on
            ref[:, :, a::2, bb::2] = F.conv2d(xp[:, :, a:a + h + 1, bb:bb + w + 1], rb(k)) + b.double().view(1, -1, 1, 1)
    close(y, ref, 3e-4, "phase forward of conv3x3(up2(x))", True)
    # 4x4 stride-2 form of sumpool2(conv3x3^T(dy)) (pack mode 1)
    dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g).to(torch.bfloat16).float()
    total = lib.ctl_conv_wpack_floats(cout, cin, 4)
    table = torch.tensor([[0, 0, cin, cout, 4, 0, 9, cin * 9, 3, 1, total, 1]], dtype=torch.int64, device=DEV)
    wp4 = torch.zeros(total, device=DEV)
    check(lib.ctl_pack_weights_bf16_batched(wt.to(DEV).contiguous().data_ptr(), wp4.data_ptr(), table.data_ptr(), 1, total, ops.stream_ptr()))
    d4 = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=cout, hout=h, wout=w, cout=cin, ks=4, stride=2, dt=dt)
    y4, _ = ops.conv_forward(d4, dev(dy, True), wp4)
    K = torch.zeros(cin, cout, 4, 4)
    for a in range(2):
        for bb in range(2):
            for kh in range(3):
                for kw in range(3):
                    K[:, :, a + 2 - kh, bb + 2 - kw] += wt[:, :, kh, kw].transpose(0, 1)
    close(y4, F.conv2d(rb(dy), rb(K), stride=2, padding=1), 3e-4, "pooled dgrad via 4x4 s2", True)


WG = [(2, 1, 16, 32, 32), (16, 4, 16, 64, 64), (2, 16, 4, 32, 32), (2, 16, 16, 32, 32), (16, 16, 16, 64, 64), (2, 32, 64, 16, 16), (2, 128, 128, 8, 8), (2, 64, 32, 24, 20), (2, 16, 32, 4, 4), (3, 48, 16, 40, 36)]


@pytest.mark.parametrize("x16,dy16", [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize("n,cin,cout,h,w", WG)
def test_bf16_wgrad_3x3(n, cin, cout, h, w, x16, dy16):
    g = torch.Generator().manual_seed(n + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    if x16:
        x = x.to(torch.bfloat16).float()
    if dy16:
        dy = dy.to(torch.bfloat16).float()
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    dt = BF | (_ffi.DT_X16 if x16 else 0) | (_ffi.DT_Y16 if dy16 else 0)
    if (x16 and cin % 16) or (dy16 and cout % 16):
        pytest.skip("bf16-stored tensors have multiples of 16 channels (network inputs / outputs are fp32)")
    for pro in (False, True):
        d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, pro_affine=int(pro), pro_slope=0.2, dt=dt)
        dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
        ops.conv_wgrad(d, dev(x, x16), dev(dy, dy16), dw, (cin * 9, 9, 3, 1), dbias=db, pro_scale=dev(sc) if pro else None,
                       pro_shift=dev(sh) if pro else None)
        xin = leaky(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2) if pro else x
        xr = rb(xin).requires_grad_(False)
        wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, wref, padding=1).backward(rb(dy))
        close(dw, wref.grad, 1e-3 if pro else 3e-4, f"wgrad 3x3 (prologue {pro})")
        close(db, rb(dy).sum((0, 2, 3)), 3e-4, "bias gradient")


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 16, 16, 16, 16), (2, 64, 32, 8, 12), (16, 16, 16, 32, 32)])
def test_bf16_wgrad_other_forms(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin * 3 + h)
    dt = BF | _ffi.DT_X16 | _ffi.DT_Y16
    # 3x3 on a nearest-upsampled input
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float()
    dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g).to(torch.bfloat16).float()
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=2 * h, wout=2 * w, cout=cout, ks=3, in_mode=_ffi.IN_UP2, dt=dt)
    dw = torch.zeros(cout, cin, 3, 3, device=DEV)
    ops.conv_wgrad(d, dev(x, True), dev(dy, True), dw, (cin * 9, 9, 3, 1))
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(F.interpolate(rb(x), scale_factor=2, mode="nearest"), wref, padding=1).backward(rb(dy))
    close(dw, wref.grad, 3e-4, "wgrad 3x3 on up2 input")
    # 3x3 stride 2
    x2 = torch.randn(n, cin, 2 * h, 2 * w, generator=g).to(torch.bfloat16).float()
    dy2 = torch.randn(n, cout, h, w, generator=g).to(torch.bfloat16).float()
    d2 = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=cin, hout=h, wout=w, cout=cout, ks=3, stride=2, dt=dt)
    dw2 = torch.zeros(cout, cin, 3, 3, device=DEV)
    ops.conv_wgrad(d2, dev(x2, True), dev(dy2, True), dw2, (cin * 9, 9, 3, 1))
    wref2 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(rb(x2), wref2, stride=2, padding=1).backward(rb(dy2))
    close(dw2, wref2.grad, 3e-4, "wgrad 3x3 stride 2")
    # 1x1 and 2x2 stride 2
    d1 = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=1, dt=dt)
    dy1 = torch.randn(n, cout, h, w, generator=g).to(torch.bfloat16).float()
    dw1 = torch.zeros(cout, cin, 1, 1, device=DEV)
    ops.conv_wgrad(d1, dev(x, True), dev(dy1, True), dw1, (cin, 1, 1, 1))
    close(dw1.view(cout, cin), torch.einsum("nchw,nkhw->kc", rb(x), rb(dy1)), 3e-4, "wgrad 1x1")
    d22 = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=cin, hout=h, wout=w, cout=cout, ks=2, stride=2, pad=0, dt=dt)
    dw22 = torch.zeros(cout, cin, 2, 2, device=DEV)
    ops.conv_wgrad(d22, dev(x2, True), dev(dy2, True), dw22, (cin * 4, 4, 2, 1))
    wref22 = torch.zeros(cout, cin, 2, 2, dtype=torch.float64, requires_grad=True)
    F.conv2d(rb(x2), wref22, stride=2).backward(rb(dy2))
    close(dw22, wref22.grad, 3e-4, "wgrad 2x2 stride 2")


# ------------------------------------------------------------------------------------------------ BatchNorm-backward prologue (round 3)
def _virtual(g, u, coef, groups, b16=True):
    """A*g + B*u + C in fp32 (coef [groups][3][c]), rounded to bf16 once in the bf16 family -- what the apply pass would have stored."""
    n = g.shape[0]
    gi = torch.arange(n) // (n // groups)
    A, B, C = (coef[gi, k].view(n, -1, 1, 1) for k in range(3))
    r = A * g + B * u + C
    return r.to(torch.bfloat16).float() if b16 else r


def _apply_on_device(g, u, coef, groups, b16=True):
    """the stand-alone apply pass (ctl_bwd_apply_dt, mode 2)"""
    n, c, h, w = g.shape
    gd, ud = dev(g, b16), dev(u, b16)
    out = torch.empty_like(gd)
    check(lib.ctl_bwd_apply_dt(2, ops.ptr(gd), None, ops.ptr(ud), None, None, 0.0, ops.ptr(dev(coef)), n * h * w, c, None, ops.ptr(out), groups,
                               (1 | 4 | 16) if b16 else 0, ops.stream_ptr()))
    return out


BNPRO = [(2, 16, 16, 32, 32, 1), (4, 32, 16, 24, 20, 2), (2, 64, 64, 16, 16, 1), (2, 128, 64, 8, 8, 2), (16, 16, 16, 128, 128, 1), (3, 48, 32, 19, 37, 1),
         (32, 16, 16, 64, 64, 2),
         # >= 8 tiles per block of the fp32 weight gradient: the pipelined (two LDS images) kernel, whole and ragged tiles, two groups
         (8, 64, 64, 64, 64, 1), (16, 64, 32, 60, 52, 2)]


@pytest.mark.parametrize("family", ["bf16", "fp32"])
@pytest.mark.parametrize("n,c,cout,h,w,groups", BNPRO)
def test_conv_bn_backward_prologue(n, c, cout, h, w, groups, family):
    """pro_affine 2: conv over the VIRTUAL tensor A*g + B*u + C (zero padding outside the image) == conv over the tensor the apply pass
    stores.  3x3 stride 1 with and without the CTL_EPI_BNBWD epilogue, and the 4x4 stride-2 pooled data gradient."""
    if n % groups:
        pytest.skip("n % groups")
    b16 = family == "bf16"
    q = (lambda t: t.to(torch.bfloat16).float()) if b16 else (lambda t: t)
    rbf = rb if b16 else (lambda t: t.double())
    pack = ops.pack_oihw_fwd_bf16 if b16 else ops.pack_oihw_fwd
    gen = torch.Generator().manual_seed(n + c + cout + h + groups)
    g = q(torch.randn(n, c, h, w, generator=gen))
    u = q(torch.randn(n, c, h, w, generator=gen))
    coef = torch.stack([torch.rand(groups, c, generator=gen) + 0.5, torch.randn(groups, c, generator=gen) * 0.3,
                        torch.randn(groups, c, generator=gen) * 0.3], 1).contiguous()            # [groups][3][c]
    virt = _virtual(g, u, coef, groups, b16)
    stored = _apply_on_device(g, u, coef, groups, b16)
    if b16:
        mism = float((stored.float().cpu() != virt).float().mean())
        assert mism < 2e-3, f"apply pass vs fp32 formula: {mism:.2e} of the elements round differently"
    else:
        close(stored, virt, 1e-6, "apply pass vs formula")
    dt = (BF | _ffi.DT_X16 | _ffi.DT_Y16) if b16 else 0
    wt = torch.randn(cout, c, 3, 3, generator=gen) * 0.2
    wp = pack(dev(wt))
    # plain
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=cout, ks=3, groups=groups, pro_affine=2, epi_flags=_ffi.EPI_STATS, dt=dt)
    # xout: the conv also writes the virtual tensor it stages (every pixel, by the tile that owns it): pre-filled with NaN to see the coverage
    xo = torch.full((n, c, h, w), float("nan"), device=DEV, dtype=torch.bfloat16 if b16 else torch.float32).contiguous(memory_format=torch.channels_last)
    y, st = ops.conv_forward(d, dev(g, b16), wp, pro_scale=dev(coef), x2=dev(u, b16), want_stats=True, xout=xo)
    ref = F.conv2d(rbf(virt), rbf(wt), padding=1)
    close(y, ref, 1e-3 if b16 else 2e-4, "conv3x3 over the virtual BatchNorm-backward tensor", b16)
    assert bool(torch.isfinite(xo.float()).all()), "xout: pixels left unwritten"
    assert torch.equal(xo.float(), stored.float()) or float((xo.float() != stored.float()).float().mean()) < 2e-3, "xout differs from the stored apply pass"
    close(xo, virt, 1e-6 if not b16 else 1e-3, "xout = the virtual tensor", b16)
    d0 = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=cout, ks=3, groups=groups, epi_flags=_ffi.EPI_STATS, dt=dt)
    y0, st0 = ops.conv_forward(d0, stored, wp, want_stats=True)
    same = float((y.float() == y0.float()).float().mean())
    assert same > (0.98 if b16 else 0.5), f"staged apply vs stored apply: only {same:.4f} of the outputs are bit-identical"
    close(y, y0.float(), 1e-3 if b16 else 1e-5, "staged apply vs stored apply")
    blocks = lib.ctl_conv_stats_blocks(_ffi.desc_ptr(d))
    close(st.view(groups, blocks, 2, cout).sum(1), st0.view(groups, -1, 2, cout).sum(1), 1e-3, "statistics")
    # with the BatchNorm-backward epilogue of the NEXT BatchNorm (the block's conv.3 data gradient)
    u1 = q(torch.randn(n, cout, h, w, generator=gen))
    sc, sh = torch.rand(groups, cout, generator=gen) + 0.5, torch.randn(groups, cout, generator=gen) * 0.3
    db = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=cout, ks=3, groups=groups, pro_affine=2,
                        epi_flags=_ffi.EPI_BNBWD | _ffi.EPI_STATS, epi_slope=0.2, dt=(dt | _ffi.DT_RES16) if b16 else 0)
    yb, stb = ops.conv_forward(db, dev(g, b16), wp, pro_scale=dev(coef), x2=dev(u, b16), res=dev(u1, b16), res_scale=dev(sc), res_shift=dev(sh),
                               want_stats=True)
    gi = torch.arange(n) // (n // groups)
    sa = u1 * sc[gi].view(n, cout, 1, 1) + sh[gi].view(n, cout, 1, 1)
    refb = ref * torch.where(sa > 0, 1.0, 0.2).double()
    close(yb, refb, 1e-3 if b16 else 2e-4, "virtual input + CTL_EPI_BNBWD", b16)
    part = stb.cpu().double().view(groups, -1, 2, cout).sum(1)
    for k in range(groups):
        sel = gi == k
        r0, r1 = refb[sel].sum((0, 2, 3)), (refb[sel] * u1[sel].double()).sum((0, 2, 3))
        assert float((part[k, 0] - r0).abs().max()) <= 2e-3 * float(r0.abs().max()) + 5e-2, "sum g"
        assert float((part[k, 1] - r1).abs().max()) <= 2e-3 * float(r1.abs().max()) + 5e-2, "sum g*u"
    # 4x4 stride 2 (the pooled data gradient of a conv on an up-sampled input)
    if h % 2 == 0 and w % 2 == 0:
        w4 = torch.randn(cout, c, 4, 4, generator=gen) * 0.2
        d4 = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h // 2, wout=w // 2, cout=cout, ks=4, stride=2, groups=groups, pro_affine=2, dt=dt)
        xo4 = torch.full_like(xo, float("nan"))
        y4, _ = ops.conv_forward(d4, dev(g, b16), pack(dev(w4)), pro_scale=dev(coef), x2=dev(u, b16), xout=xo4)
        close(y4, F.conv2d(rbf(virt), rbf(w4), stride=2, padding=1), 1e-3 if b16 else 2e-4, "conv4x4 s2 over the virtual tensor", b16)
        assert bool(torch.isfinite(xo4.float()).all()), "xout (4x4 stride 2): pixels left unwritten"
        close(xo4, virt, 1e-6 if not b16 else 1e-3, "xout of the 4x4 stride-2 form", b16)


@pytest.mark.parametrize("family", ["bf16", "fp32"])
@pytest.mark.parametrize("up", [0, 1])
@pytest.mark.parametrize("n,cin,cout,h,w,groups", BNPRO)
def test_wgrad_virtual_output_gradient(n, cin, cout, h, w, groups, up, family):
    """ctl_conv_wgrad_ex: dy = A*g + B*u + C evaluated in the staging (3x3 on a plain or nearest-up-sampled input); the bias gradient is
    the sum of the virtual tensor."""
    if n % groups or (up and (h % 2 or w % 2)):
        pytest.skip("shape")
    b16 = family == "bf16"
    q = (lambda t: t.to(torch.bfloat16).float()) if b16 else (lambda t: t)
    rbf = rb if b16 else (lambda t: t.double())
    gen = torch.Generator().manual_seed(n + cin + cout + h + groups + up)
    hx, wx = (h // 2, w // 2) if up else (h, w)
    x = q(torch.randn(n, cin, hx, wx, generator=gen))
    g = q(torch.randn(n, cout, h, w, generator=gen))
    u = q(torch.randn(n, cout, h, w, generator=gen))
    coef = torch.stack([torch.rand(groups, cout, generator=gen) + 0.5, torch.randn(groups, cout, generator=gen) * 0.3,
                        torch.randn(groups, cout, generator=gen) * 0.3], 1).contiguous()
    virt = _virtual(g, u, coef, groups, b16)
    sc, sh = torch.rand(groups, cin, generator=gen) + 0.5, torch.randn(groups, cin, generator=gen) * 0.3
    dt = (BF | _ffi.DT_X16 | _ffi.DT_Y16) if b16 else 0
    for pro in (False, True):
        d = _ffi.conv_desc(n=n, hin=hx, win=wx, cin=cin, hout=h, wout=w, cout=cout, ks=3, groups=groups, in_mode=_ffi.IN_UP2 if up else 0,
                           pro_affine=int(pro), pro_slope=0.2, dt=dt)
        dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
        ops.conv_wgrad(d, dev(x, b16), dev(g, b16), dw, (cin * 9, 9, 3, 1), dbias=db, pro_scale=dev(sc) if pro else None,
                       pro_shift=dev(sh) if pro else None, dy2=dev(u, b16), dy_coef=dev(coef))
        gi = torch.arange(n) // (n // groups)
        xin = leaky(x * sc[gi].view(n, cin, 1, 1) + sh[gi].view(n, cin, 1, 1), 0.2) if pro else x
        xr = rbf(xin)
        if up:
            xr = xr.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, wref, padding=1).backward(rbf(virt))
        close(dw, wref.grad, 1.5e-3 if b16 else 3e-4, f"wgrad with a virtual output gradient (up {up}, prologue {pro})")
        close(db, rbf(virt).sum((0, 2, 3)), 1e-3 if b16 else 3e-4, "bias gradient = sum of the virtual tensor")


@pytest.mark.parametrize("n,cin,h,w,groups", [(2, 1, 32, 32, 1), (4, 4, 24, 20, 2), (16, 1, 64, 64, 1)])
def test_wgrad_virtual_output_gradient_first_layer_fp32(n, cin, h, w, groups):
    """the K-packed <= 4-channel first layer of the fp32 family (in_mode C4) with a virtual output gradient: the encoder's inc.0 pair"""
    gen = torch.Generator().manual_seed(n + cin + h)
    cout = 16
    x = torch.randn(n, cin, h, w, generator=gen)
    g, u = torch.randn(n, cout, h, w, generator=gen), torch.randn(n, cout, h, w, generator=gen)
    coef = torch.stack([torch.rand(groups, cout, generator=gen) + 0.5, torch.randn(groups, cout, generator=gen) * 0.3,
                        torch.randn(groups, cout, generator=gen) * 0.3], 1).contiguous()
    virt = _virtual(g, u, coef, groups, False)
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, groups=groups, in_mode=_ffi.IN_C4)
    dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
    ops.conv_wgrad(d, dev(x), dev(g), dw, (cin * 9, 9, 3, 1), dbias=db, dy2=dev(u), dy_coef=dev(coef))
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wref, padding=1).backward(virt.double())
    close(dw, wref.grad, 3e-4, "first-layer wgrad with a virtual output gradient")
    close(db, virt.double().sum((0, 2, 3)), 3e-4, "bias gradient")


def test_bn_backward_prologue_argument_checks():
    n, c, h, w = 2, 16, 8, 8
    x = torch.zeros(n, c, h, w, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wp = ops.pack_oihw_fwd_bf16(torch.zeros(c, c, 3, 3, device=DEV))
    coef = torch.zeros(3 * c, device=DEV)
    dt = BF | _ffi.DT_X16 | _ffi.DT_Y16
    d = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=c, ks=3, pro_affine=2, dt=dt)
    with pytest.raises(_ffi.CtlError):
        ops.conv_forward(d, x, wp, pro_scale=coef)                                         # no x2
    d1 = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=c, ks=1, pad=0, pro_affine=2, dt=dt)
    with pytest.raises(_ffi.CtlError):
        ops.conv_forward(d1, x, ops.pack_oihw_fwd_bf16(torch.zeros(c, c, 1, 1, device=DEV)), pro_scale=coef, x2=x)      # 1x1: no such kernel
    xf = torch.zeros(n, c, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    d32 = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=c, ks=3, in_mode=_ffi.IN_UP2, pro_affine=2)
    with pytest.raises(_ffi.CtlError):
        ops.conv_forward(d32, xf, ops.pack_oihw_fwd(torch.zeros(c, c, 3, 3, device=DEV)), pro_scale=coef, x2=xf)         # up-sampled input: no such kernel
    dw = torch.zeros(c, c, 1, 1, device=DEV)
    d32w = _ffi.conv_desc(n=n, hin=h, win=w, cin=c, hout=h, wout=w, cout=c, ks=1, pad=0)
    with pytest.raises(_ffi.CtlError):
        ops.conv_wgrad(d32w, xf, xf, dw, (c, 1, 1, 1), dy2=xf, dy_coef=coef)                                           # 1x1 weight gradient: no such kernel


@pytest.mark.parametrize("family", ["fp32", "bf16", "x3"])
@pytest.mark.parametrize("form", ["1x1_accum", "1x1", "2x2_s2", "zins_3x3"])
@pytest.mark.parametrize("n,cin,cout,h,w,groups", [(2, 16, 16, 32, 32, 1), (4, 32, 16, 24, 20, 2), (2, 128, 64, 8, 8, 1), (16, 16, 16, 128, 128, 1), (3, 48, 32, 18, 38, 1)])
def test_tail_backward_epilogue(n, cin, cout, h, w, groups, form, family):
    """CTL_EPI_TAILBWD: the launch that writes dL/dOut of a residual block stores g = dOut * leaky'(out) instead and leaves the tail's
    BatchNorm-backward sums (sum g, sum g*v) in the statistics partials.  bf16 family: sums from the unrounded g, g stored as bf16."""
    if n % groups:
        pytest.skip("n % groups")
    if family == "x3" and form.startswith("1x1"):
        pytest.skip("the 1x1 hosts stay on the fp32 pipe")
    b16 = family == "bf16"
    gen = torch.Generator().manual_seed(n + cin + cout + h + len(form))
    q = (lambda t: t.to(torch.bfloat16).float()) if b16 else (lambda t: t)
    out, v = q(torch.randn(n, cout, h, w, generator=gen)), q(torch.randn(n, cout, h, w, generator=gen))
    dt = (BF | _ffi.DT_X16 | _ffi.DT_Y16 | _ffi.DT_RES16) if b16 else (_ffi.DT_X3 if family == "x3" else 0)      # (x3: the fp32 tensors, CTL_DT_X3 launch)
    flags = _ffi.EPI_TAILBWD | _ffi.EPI_STATS
    pack32 = ops.pack_oihw_fwd_x3 if family == "x3" else ops.pack_oihw_fwd
    y0 = None
    if form.startswith("1x1"):
        x = q(torch.randn(n, cin, h, w, generator=gen))
        wt = torch.randn(cout, cin, 1, 1, generator=gen) * 0.3
        wp = (ops.pack_oihw_fwd_bf16 if b16 else ops.pack_oihw_fwd)(dev(wt))
        ref = F.conv2d(rb(x) if b16 else x.double(), rb(wt) if b16 else wt.double())
        d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=1, pad=0, groups=groups, epi_slope=0.2, dt=dt,
                           epi_flags=flags | (_ffi.EPI_ACCUM if form == "1x1_accum" else 0))
        if form == "1x1_accum":
            y0 = q(torch.randn(n, cout, h, w, generator=gen))
            ref = ref + y0.double()
    elif form == "2x2_s2":
        x = q(torch.randn(n, cin, 2 * h, 2 * w, generator=gen))
        wt = torch.randn(cout, cin, 2, 2, generator=gen) * 0.3
        wp = (ops.pack_oihw_fwd_bf16 if b16 else pack32)(dev(wt))
        ref = F.conv2d(rb(x) if b16 else x.double(), rb(wt) if b16 else wt.double(), stride=2)
        d = _ffi.conv_desc(n=n, hin=2 * h, win=2 * w, cin=cin, hout=h, wout=w, cout=cout, ks=2, stride=2, pad=0, groups=groups, epi_slope=0.2, dt=dt,
                           epi_flags=flags)
    else:
        if h % 2 or w % 2:
            pytest.skip("even sizes")
        x = q(torch.randn(n, cin, h // 2, w // 2, generator=gen))
        wt = torch.randn(cout, cin, 3, 3, generator=gen) * 0.2
        wp = (ops.pack_oihw_fwd_bf16 if b16 else pack32)(dev(wt))
        xz = torch.zeros(n, cin, h, w, dtype=torch.float64)
        xz[:, :, ::2, ::2] = rb(x) if b16 else x.double()
        ref = F.conv2d(xz, rb(wt) if b16 else wt.double(), padding=1)
        d = _ffi.conv_desc(n=n, hin=h // 2, win=w // 2, cin=cin, hout=h, wout=w, cout=cout, ks=3, in_mode=_ffi.IN_ZINS2, groups=groups, epi_slope=0.2,
                           dt=dt, epi_flags=flags)
    y = dev(y0, b16) if y0 is not None else None
    # FUSE_POOL: where the tile configuration gives every wave a row pair, the 1x1 hosts also write the 2x2 sum-pool of g
    pool = None
    if form.startswith("1x1") and lib.ctl_conv_pool_ok(_ffi.desc_ptr(d)):
        pool = ops.empty_nhwc(n, cout, h // 2, w // 2, DEV, torch.bfloat16 if b16 else torch.float32)
    y, st = ops.conv_forward(d, dev(x, b16), wp, res=dev(out, b16), res2=dev(v, b16), y=y, want_stats=True, pool=pool)
    g = ref * torch.where(out > 0, 1.0, 0.2).double()
    close(y, g, 3e-4 if b16 else 2e-4, f"tail epilogue {form}", b16)
    if pool is not None:
        close(pool, F.avg_pool2d(g, 2) * 4.0, 3e-4 if b16 else 2e-4, "sum-pool of g from the tail epilogue", b16)
        if not b16:      # the stand-alone pass on the stored g: the same association, bit for bit
            sp = torch.empty_like(pool)
            check(lib.ctl_sumpool2_dt(ops.ptr(y), ops.ptr(sp), n, h // 2, w // 2, cout, 0, 0, ops.stream_ptr()))
            assert torch.equal(sp, pool), "fused sum-pool differs from ctl_sumpool2 on the stored g"
    part = st.cpu().double().view(groups, -1, 2, cout).sum(1)
    gi = torch.arange(n) // (n // groups)
    for k in range(groups):
        sel = gi == k
        r0, r1 = g[sel].sum((0, 2, 3)), (g[sel] * v[sel].double()).sum((0, 2, 3))
        assert float((part[k, 0] - r0).abs().max()) <= 5e-4 * float(g[sel].abs().sum((0, 2, 3)).max()) + 1e-3, "sum g"
        assert float((part[k, 1] - r1).abs().max()) <= 5e-4 * float((g[sel] * v[sel].double()).abs().sum((0, 2, 3)).max()) + 1e-3, "sum g*v"


def test_bf16_weight_gradients_stacked_in_one_launch():
    """ctl_conv_wgrad_group, bf16 family (round 5): weight gradients of ONE kernel instantiation stacked along blockIdx.x.  Every member keeps the
    grid and the split count of a launch of its own, so its partial sums must be BIT FOR BIT those of ctl_conv_wgrad_ex; members of another
    instantiation are refused."""
    import ctypes
    g = torch.Generator().manual_seed(5)
    dt = BF | _ffi.DT_X16 | _ffi.DT_Y16
    shapes = [(16, 32, 32, 128, 128), (2, 64, 64, 64, 64), (4, 32, 64, 64, 32), (2, 128, 128, 64, 16), (3, 64, 32, 72, 20)]      # 16x16-pixel tiles, cout tile pairs: one class
    descs, xs, dys, singles, splits, cls = [], [], [], [], [], set()
    for n, cin, cout, h, w in shapes:
        d = _ffi.conv_desc(n=n, hin=h, win=w, cin=cin, hout=h, wout=w, cout=cout, ks=3, dt=dt)
        dp = _ffi.desc_ptr(d)
        cls.add(int(lib.ctl_wgrad_group_class(dp, 0)))
        x, dy = dev(torch.randn(n, cin, h, w, generator=g), True), dev(torch.randn(n, cout, h, w, generator=g), True)
        wp = torch.full((lib.ctl_wgrad_partial_floats(dp),), float("nan"), device=DEV)
        bp = torch.full((lib.ctl_wgrad_bias_partial_floats(dp),), float("nan"), device=DEV)
        check(lib.ctl_conv_wgrad_ex(dp, x.data_ptr(), None, None, dy.data_ptr(), None, None, wp.data_ptr(), bp.data_ptr(), ops.stream_ptr()))
        descs.append(d); xs.append(x); dys.append(dy); singles.append((wp, bp)); splits.append(int(lib.ctl_wgrad_splits(dp)))
    assert len(cls) == 1 and min(cls) >= 0x100, cls
    n = len(shapes)
    darr = np.concatenate([np.atleast_1d(d) for d in descs])
    sp = np.asarray(splits, dtype=np.int32)
    wps = [torch.full_like(a, float("nan")) for a, _ in singles]
    bps = [torch.full_like(b, float("nan")) for _, b in singles]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])
    none = (ctypes.c_void_p * n)()
    check(lib.ctl_conv_wgrad_group(n, darr.ctypes.data, sp.ctypes.data, arr(xs), none, none, arr(dys), none, none, arr(wps), arr(bps), ops.stream_ptr()))
    torch.cuda.synchronize()
    for k, ((wa, ba), wb, bb) in enumerate(zip(singles, wps, bps)):
        assert torch.equal(wa, wb) and torch.equal(ba, bb), f"member {k}: the stacked launch's partial sums differ from the single launch's"
    # the planned splits: the chip's resident blocks dealt in proportion to the work -- fewer partial sums per member, the same sums
    check(lib.ctl_wgrad_group_plan(darr.ctypes.data, n, sp.ctypes.data))
    assert all(1 <= int(a) <= b for a, b in zip(sp, splits)) and int(sp.sum()) < sum(splits), (list(sp), splits)
    wps2 = [torch.full((a.numel() // s0 * int(s1),), float("nan"), device=DEV) for (a, _), s0, s1 in zip(singles, splits, sp)]
    bps2 = [torch.full((b.numel() // s0 * int(s1),), float("nan"), device=DEV) for (_, b), s0, s1 in zip(singles, splits, sp)]
    check(lib.ctl_conv_wgrad_group(n, darr.ctypes.data, sp.ctypes.data, arr(xs), none, none, arr(dys), none, none, arr(wps2), arr(bps2), ops.stream_ptr()))
    torch.cuda.synchronize()
    for k, ((wa, ba), wb, bb) in enumerate(zip(singles, wps2, bps2)):
        close(wb.view(int(sp[k]), -1).double().sum(0), wa.view(splits[k], -1).double().sum(0), 1e-5, f"member {k}: planned splits, weights")
        close(bb.view(int(sp[k]), -1).double().sum(0), ba.view(splits[k], -1).double().sum(0), 1e-5, f"member {k}: planned splits, bias")
    # a member of another instantiation (1x1) is refused
    d1 = _ffi.conv_desc(n=2, hin=64, win=64, cin=64, hout=64, wout=64, cout=64, ks=1, dt=dt)
    assert int(lib.ctl_wgrad_group_class(_ffi.desc_ptr(d1), 0)) not in cls
    bad = np.concatenate([np.atleast_1d(descs[0]), np.atleast_1d(d1)])
    sp2 = np.zeros(2, dtype=np.int32)
    check(lib.ctl_wgrad_group_plan(bad.ctypes.data, 2, sp2.ctypes.data))
    two = lambda a, b: (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    none2 = (ctypes.c_void_p * 2)()
    assert lib.ctl_conv_wgrad_group(2, bad.ctypes.data, sp2.ctypes.data, two(xs[0], xs[1]), none2, none2, two(dys[0], dys[1]), none2, none2,
                                    two(wps[0], wps[1]), two(bps[0], bps[1]), ops.stream_ptr()) != 0
    assert "class" in lib.ctl_last_error().decode()


def test_bf16_step_with_stacked_weight_gradients_equals_single_launches():
    """nets.GROUP_WGRAD_BF16: deferring the bf16 weight gradients to the end of their backward plan and stacking them changes the gradients of
    a cooperative step only by the summation order of the pixel splits (fewer splits per member): every network's gradient agrees to 1e-5
    of its largest element, and the step takes fewer launches."""
    from cooperative_training_and_latent_space_data_augmentation_amd import nets
    from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
    from oracle import ref_cpu as O
    ch = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    sp = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    c, l, nz = O.synthetic_batch(4, 64, 64, seed=11)
    prep = lambda t: (t.to(DEV).contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.to(DEV))
    c, l, nz = prep(c), prep(l), prep(nz)
    got = {}
    for on in (True, False):
        nets.GROUP_WGRAD_BF16 = on
        try:
            torch.manual_seed(0); np.random.seed(0)
            s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype="bf16")
            launches0 = lib.ctl_launch_count()
            s.cooperative_step(c, l, nz, ch, sp)
            torch.cuda.synchronize()
            got[on] = ({k: m._flat.grad.detach().clone() for k, m in s.model.items()}, lib.ctl_launch_count() - launches0)
        finally:
            nets.GROUP_WGRAD_BF16 = True
    for k in got[True][0]:
        close(got[True][0][k], got[False][0][k], 1e-5, f"gradient of {k}")
    assert got[True][1] < got[False][1], got[True][1:] + got[False][1:]
