"""CPU ORACLE for BASELINE config 3 (bf16) -- test infrastructure only, never the product path.

The reference has no bf16 path (SURVEY.md headline facts: no AMP / bf16 anywhere), so there is nothing upstream to pin a bf16
computation against.  What CAN be pinned is the engine's claim "the reference's algorithm (medseg/models/ebm/encoder_decoder.py:19-68,
285-503; BatchNorm2d semantics of model_util.py:414-451) with these ROUNDING POINTS and fp32 arithmetic everywhere else".  This file is
that statement, executable: forward AND backward of `MyEncoder` / `Dual_Branch_Encoder` / `MyDecoder` written out explicitly in torch
(fp32 or fp64 arithmetic), with a round-to-nearest-even bf16 rounding at exactly the places where the HIP plans (nets.py) store a
network-internal tensor as bf16 or feed an MFMA operand:

  forward   every conv operand (the input AFTER the fp32 BatchNorm + LeakyReLU prologue of its producer; the weights; the combined 2x2
            phase weights of conv3x3(nearest_up(x)), summed in fp32 and rounded once); every stored conv output (the BatchNorm
            statistics are taken from the UNROUNDED accumulators); the block output after the fp32 residual tail.  Network inputs /
            outputs (z, logits, images) stay fp32.
  backward  every stored gradient tensor (dS, dV, dU, the block-input gradients -- a tensor that is accumulated into is rounded at both
            stores); every dgrad / wgrad operand; the bias gradient sums the ROUNDED dy; the BatchNorm-backward sums are taken in fp32
            from the stored tensors (from the unrounded product g in the fused data-gradient epilogue, `FUSE_BNBWD16`), the
            coefficients A, B, C of dx = A*g + B*x + C in fp64.  Gradients entering / leaving a network (dz, dlogits, dx) are fp32.

With `ROUND = False` every rounding is the identity and the explicit backward below must equal autograd of the plain network: that is
how the formulas themselves are pinned (tests/test_bf16_oracle.py, CPU, fp64, 1e-9), independently of any GPU.

Works on the oracle's nn.Modules (oracle/ref_cpu.py: same parameter names as the reference).  `net_function(net)` wraps one network pass
in a torch.autograd.Function so that `OracleSolver` can run whole cooperative steps through it (`ref_cpu.bf16_rounding_points(backward=True)`)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

SLOPE = 0.2
ROUND = True                 # False: no rounding anywhere (pins the backward formulas against autograd)
FUSE_BNBWD16 = True          # nets.FUSE_BNBWD16: BatchNorm-backward sums of the in-block BatchNorm inside the data-gradient epilogue
FUSE_POOL = True             # nets.FUSE_POOL: the tail epilogue of a 1x1 host also writes sumpool2 of the UNROUNDED g (where the tile configuration allows)
FUSE_PAIR16 = True           # nets.FUSE_PAIR16: the head pairs of the encoders get g and their sums from the launch that writes dAct (CTL_EPI_BNBWD)
FUSE_TAIL16 = True           # nets.FUSE_TAIL16: the launch that writes a block's output gradient stores g = dOut * leaky'(out) and takes the tail's sums
FUSE_BNAPPLY16 = True        # nets.FUSE_BNAPPLY16: dV / dU of a residual block are never stored; their consumers apply the coefficients to the stored g


def rb(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype) if ROUND else t


def leaky(t, slope):
    return torch.where(t > 0, t, t * slope)


def dleaky(t, slope):
    """ctl_leaky_grad: derivative factor from the sign of the activation OUTPUT (== sign of its input for slope >= 0)."""
    return torch.where(t > 0, torch.ones_like(t), torch.full_like(t, slope))


def _cv(c, v):
    return v.view(1, -1, 1, 1)


# ------------------------------------------------------------------------------------------------ BatchNorm pieces
def bn_coefs(bn: nn.BatchNorm2d, u: torch.Tensor, mode: str) -> dict:
    """ctl_bn_finalize / ctl_bn_eval_coeffs: scale = gamma * invstd, shift = beta - mean * scale from the UNROUNDED conv output `u`
    (sums in fp64); mode 'A' also updates the running statistics (momentum, unbiased variance), 'C' uses them."""
    dt = u.dtype
    if mode == "C":
        sc = bn.weight.detach().to(dt) / torch.sqrt(bn.running_var.to(dt) + bn.eps)
        return {"scale": sc, "shift": bn.bias.detach().to(dt) - bn.running_mean.to(dt) * sc, "mean": None, "invstd": None}
    cnt = u.numel() // u.shape[1]
    u64 = u.double()
    mean = u64.mean((0, 2, 3))
    var = (u64 * u64).mean((0, 2, 3)) - mean * mean
    var = var.clamp_min(0.0)
    invstd = (1.0 / torch.sqrt(var + bn.eps)).to(dt)
    sc = bn.weight.detach().to(dt) * invstd
    sh = bn.bias.detach().to(dt) - mean.to(dt) * sc
    if mode == "A":
        with torch.no_grad():
            unbiased = var * cnt / (cnt - 1) if cnt > 1 else var
            bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * mean.to(bn.running_mean.dtype))
            bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * unbiased.to(bn.running_var.dtype))
            bn.num_batches_tracked.add_(1)
    return {"scale": sc, "shift": sh, "mean": mean.to(dt), "invstd": invstd}


def bn_bwd_coefs(bn: nn.BatchNorm2d, co: dict, s1: torch.Tensor, s2: torch.Tensor, cnt: int, dt):
    """bn_bwd_coefs (ctl_elem.hip): dx = A*g + B*x + C, dgamma = sum g*xhat, dbeta = sum g; fp64 from the two sums."""
    mu, is_, g = co["mean"].double(), co["invstd"].double(), bn.weight.detach().double()
    sum_g, sum_gx = s1.double(), is_ * (s2.double() - mu * s1.double())
    m1, m2 = sum_g / cnt, sum_gx / cnt
    A = g * is_
    B = -g * is_ * is_ * m2
    C = -g * is_ * m1 + g * is_ * is_ * m2 * mu
    return A.to(dt), B.to(dt), C.to(dt), sum_gx.to(dt), sum_g.to(dt)


def bn_backward(bn, co, g: torch.Tensor, x_st: torch.Tensor):
    """reduce -> finalize -> apply on g (the gradient already multiplied by the activation derivative) and the stored BatchNorm input:
    returns (dx unrounded, dgamma, dbeta)."""
    cnt = g.numel() // g.shape[1]
    s1, s2 = g.sum((0, 2, 3)), (g * x_st).sum((0, 2, 3))
    A, B, C, dgamma, dbeta = bn_bwd_coefs(bn, co, s1, s2, cnt, g.dtype)
    return _cv(0, A) * g + _cv(0, B) * x_st + _cv(0, C), dgamma, dbeta


# ------------------------------------------------------------------------------------------------ conv pieces
def conv_fwd(conv: nn.Conv2d, x_st: torch.Tensor, pro: Optional[Tuple] = None) -> torch.Tensor:
    """ctl_conv_forward: operand = bf16(prologue(x)), weights bf16, fp32 accumulate + bias; result UNROUNDED."""
    a = x_st if pro is None else leaky(x_st * _cv(0, pro[0]) + _cv(0, pro[1]), pro[2])
    return F.conv2d(rb(a), rb(conv.weight.detach().to(a.dtype)), conv.bias.detach().to(a.dtype), conv.stride, conv.padding)


def conv_operand(x_st, pro):
    return rb(x_st if pro is None else leaky(x_st * _cv(0, pro[0]) + _cv(0, pro[1]), pro[2]))


def conv_dgrad(conv: nn.Conv2d, dy_st: torch.Tensor, in_hw: Tuple[int, int]) -> torch.Tensor:
    """data gradient: conv over bf16(dy) with the transposed / flipped bf16 weights; result UNROUNDED."""
    s, p, k = conv.stride[0], conv.padding[0], conv.kernel_size[0]
    oph = in_hw[0] - ((dy_st.shape[2] - 1) * s - 2 * p + k)
    opw = in_hw[1] - ((dy_st.shape[3] - 1) * s - 2 * p + k)
    return F.conv_transpose2d(rb(dy_st), rb(conv.weight.detach().to(dy_st.dtype)), None, stride=s, padding=p, output_padding=(oph, opw))


def conv_wgrad(conv: nn.Conv2d, x_op: torch.Tensor, dy_st: torch.Tensor):
    """weight gradient from the bf16 operands (x_op is already the rounded operand), fp32 accumulate; bias gradient = sum of bf16(dy)."""
    dy = rb(dy_st)
    dw = torch.nn.grad.conv2d_weight(x_op, conv.weight.shape, dy, stride=conv.stride, padding=conv.padding)
    return dw, dy.sum((0, 2, 3))


def up2(x):
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def sumpool2(x):
    n, c, h, w = x.shape
    x = x.view(n, c, h // 2, 2, w // 2, 2)
    return (x[:, :, :, 0, :, 0] + x[:, :, :, 0, :, 1]) + (x[:, :, :, 1, :, 0] + x[:, :, :, 1, :, 1])


def phase_kernels(w: torch.Tensor):
    """conv3x3(nearest_up(x)) as four 2x2 convs on x (pack mode 2): the 3x3 taps that land on the same source pixel are summed in fp32
    and the combined kernel is rounded once.  Returns {(a, b): kernel [co, ci, 2, 2]} for output phase (2i+a, 2j+b)."""
    out = {}
    for a in range(2):
        for b in range(2):
            k = w.new_zeros(w.shape[0], w.shape[1], 2, 2)
            for kh in range(3):
                for kw in range(3):
                    k[:, :, (a + kh + 1) // 2 - a, (b + kw + 1) // 2 - b] += w[:, :, kh, kw]
            out[(a, b)] = k
    return out


def conv3x3_on_up2(conv: nn.Conv2d, x_st: torch.Tensor) -> torch.Tensor:
    n, _, h, w = x_st.shape
    wt = conv.weight.detach().to(x_st.dtype)
    xp = F.pad(rb(x_st), (1, 1, 1, 1))
    u = x_st.new_zeros(n, wt.shape[0], 2 * h, 2 * w)
    for (a, b), k in phase_kernels(wt).items():
        u[:, :, a::2, b::2] = F.conv2d(xp[:, :, a:a + h + 1, b:b + w + 1], rb(k), conv.bias.detach().to(x_st.dtype))
    return u


def pooled_dgrad_kernel(w: torch.Tensor) -> torch.Tensor:
    """sumpool2(conv3x3^T(dU)) as ONE 4x4 stride-2 pad-1 conv over dU (pack mode 1): K[u] = sum over a in {0,1} of W[a + 2 - u] (taps
    inside [0, 2]), summed in fp32, rounded once.  Returned as a conv2d weight [ci, co, 4, 4]."""
    k = w.new_zeros(w.shape[0], w.shape[1], 4, 4)
    for uh in range(4):
        for a in range(2):
            sh = a + 2 - uh
            if sh < 0 or sh > 2:
                continue
            for uw in range(4):
                for b in range(2):
                    sw = b + 2 - uw
                    if sw < 0 or sw > 2:
                        continue
                    k[:, :, uh, uw] += w[:, :, sh, sw]
    return k.permute(1, 0, 2, 3).contiguous()


# ------------------------------------------------------------------------------------------------ residual blocks
def block_fwd(blk, pre: str, xin: torch.Tensor, xin_pro, mode: str) -> Tuple[torch.Tensor, dict]:
    """nets._emit_block_fwd: res_convdown (pre='down') / res_up_family (pre='nn' | 'convT').  `xin` is what the producer stored (bf16
    values, or the fp32 network input); returns the stored block output and the record the backward needs."""
    c0, bn1, _, c3, bn2 = blk.conv[0], blk.conv[1], blk.conv[2], blk.conv[3], blk.conv[4]
    c1 = blk.conv_input
    rec = {"pre": pre, "xin": xin, "xin_pro": xin_pro}
    if pre == "down":
        src = rb(conv_fwd(blk.down, xin, xin_pro))
        u_raw = conv_fwd(c0, src)
    elif pre == "convT":
        t = blk.up
        src = rb(F.conv_transpose2d(rb(xin), rb(t.weight.detach().to(xin.dtype)), t.bias.detach().to(xin.dtype), stride=2))
        u_raw = conv_fwd(c0, src)
    else:
        src = xin                                         # virtual: nearest-upsampled while staging
        u_raw = conv3x3_on_up2(c0, xin)
    co1 = bn_coefs(bn1, u_raw, mode)
    u = rb(u_raw)
    v_raw = conv_fwd(c3, u, (co1["scale"], co1["shift"], SLOPE))
    co2 = bn_coefs(bn2, v_raw, mode)
    v = rb(v_raw)
    s_op = rb(up2(src)) if pre == "nn" else rb(src)
    s_raw = F.conv2d(s_op, rb(c1.weight.detach().to(src.dtype)), c1.bias.detach().to(src.dtype))
    out = rb(leaky(s_raw + (v * _cv(0, co2["scale"]) + _cv(0, co2["shift"])), SLOPE))
    rec.update(src=src, u=u, v=v, out=out, co1=co1, co2=co2)
    return out, rec


def tail_pack(t: torch.Tensor, tail_next):
    """CTL_EPI_TAILBWD epilogue of the launch that writes the output gradient of the block (out, v) = tail_next: g = dOut * leaky'(out)
    from the UNROUNDED dOut = t, the tail's BatchNorm-backward sums from the unrounded g, g stored.  Returns (g stored, (sum g, sum g*v))."""
    out_n, v_n = tail_next[:2]
    g = t * dleaky(out_n, SLOPE)
    sums = (g.sum((0, 2, 3)), (g * v_n).sum((0, 2, 3)))
    if len(tail_next) > 2 and tail_next[2]:          # ... and the 2x2 sum-pool of the unrounded g, stored (nets.FUSE_POOL)
        return rb(g), sums + (rb(sumpool2(g)),)
    return rb(g), sums


def act_pack(t: torch.Tensor, act_next):
    """CTL_EPI_BNBWD epilogue of the launch that writes dAct of a conv-BatchNorm-activation pair, act_next = (u, co, slope): g = dAct *
    act'(BN(u)) from the UNROUNDED dAct = t, sums from the unrounded g, g stored.  Returns (g stored, (sum g, sum g*u))."""
    u_n, co_n, slope = act_next
    g = t * dleaky(u_n * _cv(0, co_n["scale"]) + _cv(0, co_n["shift"]), slope)
    return rb(g), (g.sum((0, 2, 3)), (g * u_n).sum((0, 2, 3)))


def tail_of(brec, pooled=None):
    """pooled: does the producing launch also write the sum-pool (None = whenever the consumer is a nearest-upsample block: what the plans do
    where the producer's tile configuration gives every wave a row pair, i.e. everywhere but on the smallest layers)"""
    if not FUSE_TAIL16:
        return None
    return (brec["out"], brec["v"], (FUSE_POOL and brec["pre"] == "nn") if pooled is None else pooled)


def block_bwd(blk, rec: dict, d_out: torch.Tensor, need_w: bool, affine: bool, grads: dict, prefix: str, last: bool, *, pre_tail=None,
              d_out_is_g: bool = False, tail_next=None, act_next=None):
    """nets._emit_block_bwd.  d_out: gradient w.r.t. the block output as stored by its producer.  Returns the gradient w.r.t. the block
    input (w.r.t. the activated virtual tensor if rec['xin_pro'] is set): rounded like a stored bf16 tensor unless `last` (then it is the
    fp32 gradient leaving the network).
    pre_tail = (sum g, sum g*v) from `tail_pack`: d_out is already the stored g of this block's tail (d_out_is_g without sums: teacher-
    forced tests, the sums are then taken from the stored g).  tail_next = (out, v) of the block that consumes this block's input
    gradient: returns `tail_pack` of it instead; act_next = (u, co, slope) of the pair that consumes it: `act_pack`."""
    c0, bn1, c3, bn2, c1 = blk.conv[0], blk.conv[1], blk.conv[3], blk.conv[4], blk.conv_input
    pre, src, u, v, out, xin = rec["pre"], rec["src"], rec["u"], rec["v"], rec["out"], rec["xin"]
    co1, co2 = rec["co1"], rec["co2"]
    assert not (last and (tail_next is not None or act_next is not None)) and not (tail_next is not None and act_next is not None)
    store = (lambda t: t) if last else ((lambda t: tail_pack(t, tail_next)) if tail_next is not None else
                                        ((lambda t: act_pack(t, act_next)) if act_next is not None else rb))
    # residual tail (BatchNorm-backward mode 0): g = dOut * leaky'(out); dS = g, dV = A*g + B*v + C
    if pre_tail is not None or d_out_is_g:
        ds = d_out
        s1, s2 = pre_tail[:2] if pre_tail is not None else (ds.sum((0, 2, 3)), (ds * v).sum((0, 2, 3)))
        A2, B2, C2, dg2, db2 = bn_bwd_coefs(bn2, co2, s1, s2, ds.numel() // ds.shape[1], ds.dtype)
        dv = rb(_cv(0, A2) * ds + _cv(0, B2) * v + _cv(0, C2))          # (apply pass on the stored g, stand-alone or staged: same arithmetic)
        g = None
    else:
        g = d_out * dleaky(out, SLOPE)
    if g is None:
        pass
    elif FUSE_BNAPPLY16:
        # no apply pass: the reduction stores dS = bf16(g) (sums from the unrounded g), and the consumers of dV evaluate A*dS + B*v + C
        # on the STORED dS while staging, rounding the result once to their bf16 operand
        cnt = g.numel() // g.shape[1]
        A2, B2, C2, dg2, db2 = bn_bwd_coefs(bn2, co2, g.sum((0, 2, 3)), (g * v).sum((0, 2, 3)), cnt, g.dtype)
        ds = rb(g)
        dv = rb(_cv(0, A2) * ds + _cv(0, B2) * v + _cv(0, C2))
    else:
        dv_raw, dg2, db2 = bn_backward(bn2, co2, g, v)
        ds, dv = rb(g), rb(dv_raw)
    if need_w and affine:
        grads[prefix + ".conv.4.weight"], grads[prefix + ".conv.4.bias"] = dg2, db2
    pro1 = (co1["scale"], co1["shift"], SLOPE)
    if need_w:
        grads[prefix + ".conv.3.weight"], grads[prefix + ".conv.3.bias"] = conv_wgrad(c3, conv_operand(u, pro1), dv)
    da = conv_dgrad(c3, dv, u.shape[2:])
    sa = u * _cv(0, co1["scale"]) + _cv(0, co1["shift"])
    if FUSE_BNBWD16:
        # data-gradient epilogue: g1 = dA * leaky'(BN1(u)), sums from the UNROUNDED g1, g1 stored; apply on the stored g1 (in place)
        g1 = da * dleaky(sa, SLOPE)
        cnt = g1.numel() // g1.shape[1]
        A, B, C, dg1, db1 = bn_bwd_coefs(bn1, co1, g1.sum((0, 2, 3)), (g1 * u).sum((0, 2, 3)), cnt, g1.dtype)
        du = rb(_cv(0, A) * rb(g1) + _cv(0, B) * u + _cv(0, C))
    else:
        g1 = rb(da) * dleaky(sa, SLOPE)
        du_raw, dg1, db1 = bn_backward(bn1, co1, g1, u)
        du = rb(du_raw)
    if need_w and affine:
        grads[prefix + ".conv.1.weight"], grads[prefix + ".conv.1.bias"] = dg1, db1
    if pre == "nn":
        ds_low = pre_tail[2] if (pre_tail is not None and len(pre_tail) > 2) else rb(sumpool2(ds))
        if need_w:
            grads[prefix + ".conv.0.weight"], grads[prefix + ".conv.0.bias"] = conv_wgrad(c0, up2(rb(xin)), du)
            grads[prefix + ".conv_input.weight"], grads[prefix + ".conv_input.bias"] = conv_wgrad(c1, rb(xin), ds_low)
        d1 = F.conv2d(du, rb(pooled_dgrad_kernel(c0.weight.detach().to(du.dtype))), None, stride=2, padding=1)
        d1 = d1 if last else rb(d1)                                      # first launch stores, the second accumulates onto the stored tensor
        return store(d1 + conv_dgrad(c1, ds_low, xin.shape[2:]))
    if need_w:
        grads[prefix + ".conv.0.weight"], grads[prefix + ".conv.0.bias"] = conv_wgrad(c0, rb(src), du)
        grads[prefix + ".conv_input.weight"], grads[prefix + ".conv_input.bias"] = conv_wgrad(c1, rb(src), ds)
    dsrc = rb(conv_dgrad(c0, du, src.shape[2:]))
    dsrc = rb(dsrc + conv_dgrad(c1, ds, src.shape[2:]))
    if pre == "convT":
        t = blk.up
        if need_w:
            grads[prefix + ".up.bias"] = dsrc.sum((0, 2, 3))
            # role swap: W_t[ci][co][a][b] = sum_pixels x[i,j,ci] * dsrc[2i+a, 2j+b, co]
            grads[prefix + ".up.weight"] = torch.nn.grad.conv2d_weight(dsrc, (t.weight.shape[0], t.weight.shape[1], 2, 2), rb(xin), stride=2)
        return store(F.conv2d(dsrc, rb(t.weight.detach().to(dsrc.dtype)), None, stride=2))
    if need_w:
        grads[prefix + ".down.weight"], grads[prefix + ".down.bias"] = conv_wgrad(blk.down, conv_operand(xin, rec["xin_pro"]), dsrc)
    return store(conv_dgrad(blk.down, dsrc, xin.shape[2:]))


def conv_bn_pair_bwd(conv, bn, x, x_pro, u, co, slope, d_act, need_w, affine, grads, ckey, bkey, need_dx=True, last=False, tail_next=None,
                     pre=None, act_next=None):
    """nets._emit_conv_bn_pair_bwd: backward of a = act(BN(conv(x))) given d_act (as stored); returns the gradient w.r.t. x
    (`tail_pack` / `act_pack` of it if tail_next / act_next is given).  pre = (sum g, sum g*u): d_act is the stored g of `act_pack`."""
    if pre is not None:
        A, B, C, dg, db = bn_bwd_coefs(bn, co, pre[0], pre[1], d_act.numel() // d_act.shape[1], d_act.dtype)
        du = rb(_cv(0, A) * d_act + _cv(0, B) * u + _cv(0, C))          # apply on the stored g (staged in the consumers or stand-alone: same arithmetic)
    else:
        g = d_act * dleaky(u * _cv(0, co["scale"]) + _cv(0, co["shift"]), slope)
        du_raw, dg, db = bn_backward(bn, co, g, u)
        du = rb(du_raw)
    if need_w and affine:
        grads[bkey + ".weight"], grads[bkey + ".bias"] = dg, db
    if need_w:
        grads[ckey + ".weight"], grads[ckey + ".bias"] = conv_wgrad(conv, conv_operand(x, x_pro), du)
    if not need_dx:
        return None
    dx = conv_dgrad(conv, du, x.shape[2:])
    if tail_next is not None:
        return tail_pack(dx, tail_next)
    if act_next is not None:
        return act_pack(dx, act_next)
    return dx if last else rb(dx)


# ------------------------------------------------------------------------------------------------ networks
def net_mode(net: nn.Module) -> str:
    if not net.training:
        return "C"
    bn = next(m for m in net.modules() if isinstance(m, nn.BatchNorm2d))
    return "A" if bn.track_running_stats else "B"


def encoder_fwd(enc, x: torch.Tensor, mode: str, px: str = ""):
    c0, b0, _, c3, b3 = enc.inc[0], enc.inc[1], enc.inc[2], enc.inc[3], enc.inc[4]
    u0_raw = conv_fwd(c0, x)
    co0 = bn_coefs(b0, u0_raw, mode)
    u0 = rb(u0_raw)
    v0_raw = conv_fwd(c3, u0, (co0["scale"], co0["shift"], SLOPE))
    co1 = bn_coefs(b3, v0_raw, mode)
    v0 = rb(v0_raw)
    rec = {"x": x, "u0": u0, "v0": v0, "co0": co0, "co1": co1, "blocks": [], "px": px}
    cur, cur_pro = v0, (co1["scale"], co1["shift"], SLOPE)
    for i in range(1, 5):
        cur, brec = block_fwd(getattr(enc, f"down{i}"), "down", cur, cur_pro, mode)
        cur_pro = None
        rec["blocks"].append(brec)
    cf, bf = enc.final_conv[0], enc.final_conv[1]
    uf_raw = conv_fwd(cf, cur)
    cof = bn_coefs(bf, uf_raw, mode)
    uf = rb(uf_raw)
    z = leaky(uf * _cv(0, cof["scale"]) + _cv(0, cof["shift"]), 0.0)          # ReLU; the network output stays fp32
    rec.update(uf=uf, cof=cof, x4=cur, z=z)
    return z, rec


def encoder_bwd(enc, rec, dz, need_dx, need_w, affine, grads):
    px = rec["px"]
    blocks, pre = rec["blocks"], None
    d = conv_bn_pair_bwd(enc.final_conv[0], enc.final_conv[1], rec["x4"], None, rec["uf"], rec["cof"], 0.0, dz, need_w, affine, grads,
                         px + "final_conv.0", px + "final_conv.1", tail_next=tail_of(blocks[3]))
    if FUSE_TAIL16:
        d, pre = d
    pair = FUSE_PAIR16 and FUSE_BNAPPLY16
    act1 = (rec["v0"], rec["co1"], SLOPE) if pair else None
    act0 = (rec["u0"], rec["co0"], SLOPE) if pair else None
    for i in range(4, 0, -1):
        tn = tail_of(blocks[i - 2]) if i >= 2 else None
        an = act1 if i == 1 else None
        d = block_bwd(getattr(enc, f"down{i}"), blocks[i - 1], d, need_w, affine, grads, f"{px}down{i}", last=False, pre_tail=pre, tail_next=tn, act_next=an)
        d, pre = d if (tn is not None or an is not None) else (d, None)
    pro0 = (rec["co0"]["scale"], rec["co0"]["shift"], SLOPE)
    d = conv_bn_pair_bwd(enc.inc[3], enc.inc[4], rec["u0"], pro0, rec["v0"], rec["co1"], SLOPE, d, need_w, affine, grads, px + "inc.3", px + "inc.4",
                         pre=pre, act_next=act0)
    d, pre = d if act0 is not None else (d, None)
    return conv_bn_pair_bwd(enc.inc[0], enc.inc[1], rec["x"], None, rec["u0"], rec["co0"], SLOPE, d, need_w, affine, grads, px + "inc.0", px + "inc.1",
                            need_dx=need_dx, last=True, pre=pre)


def dual_fwd(net, x, mode):
    z_i, rec = encoder_fwd(net.general_encoder, x, mode, "general_encoder.")
    d0, b0, _, d3, b3 = net.code_decoupler[0], net.code_decoupler[1], net.code_decoupler[2], net.code_decoupler[3], net.code_decoupler[4]
    ud_raw = conv_fwd(d0, z_i)
    cod0 = bn_coefs(b0, ud_raw, mode)
    ud = rb(ud_raw)
    vd_raw = conv_fwd(d3, ud, (cod0["scale"], cod0["shift"], SLOPE))
    cod1 = bn_coefs(b3, vd_raw, mode)
    vd = rb(vd_raw)
    z_s = leaky(vd * _cv(0, cod1["scale"]) + _cv(0, cod1["shift"]), 0.0)
    rec.update(ud=ud, vd=vd, cod0=cod0, cod1=cod1, z_i=z_i)
    return (z_i, z_s), rec


def dual_bwd(net, rec, dzi_in, dzs, need_dx, need_w, affine, grads):
    d0, b0, d3, b3 = net.code_decoupler[0], net.code_decoupler[1], net.code_decoupler[3], net.code_decoupler[4]
    dzi = dzi_in
    if dzs is not None:
        pro = (rec["cod0"]["scale"], rec["cod0"]["shift"], SLOPE)
        d = conv_bn_pair_bwd(d3, b3, rec["ud"], pro, rec["vd"], rec["cod1"], 0.0, dzs, need_w, affine, grads, "code_decoupler.3", "code_decoupler.4")
        g = d * dleaky(rec["ud"] * _cv(0, rec["cod0"]["scale"]) + _cv(0, rec["cod0"]["shift"]), SLOPE)
        du_raw, dg, db = bn_backward(b0, rec["cod0"], g, rec["ud"])
        du = rb(du_raw)
        if need_w and affine:
            grads["code_decoupler.1.weight"], grads["code_decoupler.1.bias"] = dg, db
        if need_w:
            grads["code_decoupler.0.weight"], grads["code_decoupler.0.bias"] = conv_wgrad(d0, rb(rec["z_i"]), du)
        dd = conv_dgrad(d0, du, rec["z_i"].shape[2:])                      # accumulated into the fp32 dz_i: no rounding
        dzi = dd if dzi_in is None else dzi_in + dd
    return encoder_bwd(net.general_encoder, rec, dzi, need_dx, need_w, affine, grads)


def decoder_fwd(dec, z, mode):
    pre = "nn" if isinstance(dec.up1.up, nn.Sequential) else "convT"
    cur, rec = z, {"x": z, "blocks": [], "pre": pre}
    for i in range(1, 5):
        cur, brec = block_fwd(getattr(dec, f"up{i}"), pre, cur, None, mode)
        rec["blocks"].append(brec)
    cf = dec.final_conv
    out = F.conv2d(rb(cur), rb(cf.weight.detach().to(cur.dtype)), cf.bias.detach().to(cur.dtype))
    if dec.last_act is not None:
        out = torch.sigmoid(out)
    rec.update(x4=cur, out=out)
    return out, rec


def decoder_bwd(dec, rec, dout, need_dx, need_w, affine, grads):
    cf = dec.final_conv
    if dec.last_act is not None:
        dout = dout * rec["out"] * (1 - rec["out"])
    if need_w:
        grads["final_conv.weight"], grads["final_conv.bias"] = conv_wgrad(cf, rb(rec["x4"]), dout)
    blocks, pre = rec["blocks"], None
    t = conv_dgrad(cf, dout, rec["x4"].shape[2:])
    d, pre = tail_pack(t, tail_of(blocks[3])) if FUSE_TAIL16 else (rb(t), None)
    for i in range(4, 0, -1):
        tn = tail_of(blocks[i - 2]) if i >= 2 else None
        d = block_bwd(getattr(dec, f"up{i}"), blocks[i - 1], d, need_w, affine, grads, f"up{i}", last=(i == 1), pre_tail=pre, tail_next=tn)
        d, pre = d if tn is not None else (d, None)
    return d if need_dx else None


def net_forward(net, x, mode=None):
    """(outputs tuple, rec) of one pass of an oracle network (Encoder / DualEncoder / Decoder of ref_cpu.py) in the engine's bf16 arithmetic."""
    from . import ref_cpu as O
    mode = mode or net_mode(net)
    with torch.no_grad():
        if isinstance(net, O.DualEncoder):
            outs, rec = dual_fwd(net, x, mode)
        elif isinstance(net, O.Encoder):
            z, rec = encoder_fwd(net, x, mode)
            if net.act is None:
                raise NotImplementedError("the path's encoders end in ReLU")
            outs = (z,)
        else:
            y, rec = decoder_fwd(net, x, mode)
            outs = (y,)
    rec["mode"] = mode
    return outs, rec


def net_backward(net, rec, douts, need_dx=True, need_w=True):
    """(dx, {parameter name: gradient}) of one pass; douts aligned with the outputs of net_forward (None = no gradient)."""
    from . import ref_cpu as O
    grads: Dict[str, torch.Tensor] = {}
    affine = rec["mode"] == "A"
    with torch.no_grad():
        if isinstance(net, O.DualEncoder):
            dx = dual_bwd(net, rec, douts[0], douts[1] if len(douts) > 1 else None, need_dx, need_w, affine, grads)
        elif isinstance(net, O.Encoder):
            dx = encoder_bwd(net, rec, douts[0], need_dx, need_w, affine, grads)
        else:
            dx = decoder_bwd(net, rec, douts[0], need_dx, need_w, affine, grads)
    return dx, grads


class _NetFnB16(torch.autograd.Function):
    """One network pass as an autograd node: forward / backward are the explicit bf16-plan computations above."""

    @staticmethod
    def forward(ctx, net, x, *params):
        outs, rec = net_forward(net, x.detach())
        ctx.net, ctx.rec = net, rec
        ctx.names = [n for n, _ in net.named_parameters()]
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        net, rec = ctx.net, ctx.rec
        if rec["mode"] == "C":
            raise RuntimeError("backward through an eval-mode pass is not part of the hot path")
        need_dx = ctx.needs_input_grad[1]
        need_w = any(ctx.needs_input_grad[2:])
        douts = [None if d is None else d.detach() for d in douts]
        dx, grads = net_backward(net, rec, douts, need_dx, need_w)
        out = [None, dx]
        for i, n in enumerate(ctx.names):
            g = grads.get(n) if ctx.needs_input_grad[2 + i] else None
            out.append(g)
        return tuple(out)


def net_apply(net, x):
    """`net(x)` through the bf16-plan emulation (autograd-aware)."""
    outs = _NetFnB16.apply(net, x, *[p for _, p in net.named_parameters()])
    from . import ref_cpu as O
    return outs if isinstance(net, O.DualEncoder) else outs[0]
