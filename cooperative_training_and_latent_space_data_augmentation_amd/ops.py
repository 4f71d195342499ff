"""Thin tensor-level wrappers over the C-ABI (one call = one kernel enqueue on torch's current stream).

PyTorch is used here only for device memory and the stream handle.  4-D activations are logical NCHW tensors in
``torch.channels_last`` memory format, i.e. NHWC in HBM, which is what every kernel expects.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _ffi
from ._ffi import lib, check


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _ffi.CtlError("the HIP path needs device tensors (got a CPU tensor); there is no CPU fallback")


def as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor whose memory is NHWC (no copy if it already is)."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor")
    return x.float().contiguous(memory_format=torch.channels_last)


def empty_nhwc(n: int, c: int, h: int, w: int, device, dtype=torch.float32) -> torch.Tensor:
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


def as_nhwc_any(x: torch.Tensor) -> torch.Tensor:
    """NHWC memory, dtype kept (fp32 or bf16): the bf16 kernel family reads either (ctl_conv.dt)."""
    if x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("expected a 4-D fp32 / bf16 NCHW tensor")
    return x.contiguous(memory_format=torch.channels_last)


# ---------------------------------------------------------------------------------------------- conv (unit-level API)
def pack_weights(src: torch.Tensor, cout: int, cin: int, ks: int, strides, flip: bool, src_offset: int = 0) -> torch.Tensor:
    require_gpu(src)
    n = lib.ctl_conv_wpack_floats(cin, cout, ks)
    dst = torch.empty(n, dtype=torch.float32, device=src.device)
    check(lib.ctl_pack_weights(src.data_ptr() + 4 * src_offset, dst.data_ptr(), cout, cin, ks, *[int(s) for s in strides],
                               int(flip), stream_ptr()), "ctl_pack_weights")
    return dst


def pack_oihw_fwd(w: torch.Tensor) -> torch.Tensor:
    co, ci, ks, _ = w.shape
    return pack_weights(w.contiguous(), co, ci, ks, (ci * ks * ks, ks * ks, ks, 1), False)


def pack_oihw_dgrad(w: torch.Tensor) -> torch.Tensor:
    co, ci, ks, _ = w.shape
    return pack_weights(w.contiguous(), ci, co, ks, (ks * ks, ci * ks * ks, ks, 1), True)


def pack_weights_bf16(w: torch.Tensor, cout: int, cin: int, ks: int, strides, flip=0, mode: int = 0) -> torch.Tensor:
    """bf16 MFMA fragments (tap pairs x 16-channel chunks, see ctl_conv_bf16.hip) of one effective conv through the table-driven pack
    kernel; `strides`/`flip`/`mode` as in ctl_pack_weights_batched records."""
    require_gpu(w)
    total = lib.ctl_conv_wpack_floats(cin, cout, ks)
    table = torch.tensor([[0, 0, cout, cin, ks, int(flip), *[int(v) for v in strides], total, mode]], dtype=torch.int64, device=w.device)
    dst = torch.zeros(total, dtype=torch.float32, device=w.device)
    src = w.contiguous().float()
    check(lib.ctl_pack_weights_bf16_batched(src.data_ptr(), dst.data_ptr(), table.data_ptr(), 1, total, stream_ptr()), "ctl_pack_weights_bf16_batched")
    return dst


def pack_weights_x3(w: torch.Tensor, cout: int, cin: int, ks: int, strides, flip=0, mode: int = 0) -> torch.Tensor:
    """Three-plane bf16 fragments for CTL_DT_X3 launches (hi | mid | lo planes summing exactly to the fp32 weight, ctl_conv_x3.hip) of
    one effective conv; `strides` / `flip` / `mode` as in ctl_pack_weights_batched records."""
    require_gpu(w)
    total = lib.ctl_conv_wpack_floats_x3(cin, cout, ks)
    table = torch.tensor([[0, 0, cout, cin, ks, int(flip), *[int(v) for v in strides], total, mode | _ffi.PACK_X3]], dtype=torch.int64, device=w.device)
    dst = torch.zeros(total, dtype=torch.float32, device=w.device)
    src = w.contiguous().float()
    check(lib.ctl_pack_weights_x3_batched(src.data_ptr(), dst.data_ptr(), table.data_ptr(), 1, total, stream_ptr()), "ctl_pack_weights_x3_batched")
    return dst


def pack_oihw_fwd_x3(w: torch.Tensor) -> torch.Tensor:
    co, ci, ks, _ = w.shape
    return pack_weights_x3(w, co, ci, ks, (ci * ks * ks, ks * ks, ks, 1), 0)


def pack_oihw_dgrad_x3(w: torch.Tensor) -> torch.Tensor:
    co, ci, ks, _ = w.shape
    return pack_weights_x3(w, ci, co, ks, (ks * ks, ci * ks * ks, ks, 1), 1)


def pack_oihw_fwd_bf16(w: torch.Tensor) -> torch.Tensor:
    co, ci, ks, _ = w.shape
    return pack_weights_bf16(w, co, ci, ks, (ci * ks * ks, ks * ks, ks, 1), 0)


def pack_oihw_dgrad_bf16(w: torch.Tensor) -> torch.Tensor:
    co, ci, ks, _ = w.shape
    return pack_weights_bf16(w, ci, co, ks, (ks * ks, ci * ks * ks, ks, 1), 1)


def conv_forward(d: np.ndarray, x, wpack, bias=None, pro_scale=None, pro_shift=None, res=None, res_scale=None,
                 res_shift=None, y=None, want_stats=False, x2=None, res2=None, pool=None, xout=None):
    """Run one conv problem described by the ctl_conv record `d`.  Returns (y, stats_partial or None).  With d["dt"] & DT_BF16 the
    storage dtypes of x / res / y must agree with the DT_X16 / DT_RES16 / DT_Y16 flags (y is allocated accordingly).
    x2 (d["pro_affine"] == 2): the BatchNorm-backward prologue, input = A*x + B*x2 + C with pro_scale = [groups][3][cin] coefficients.
    res2 (CTL_EPI_TAILBWD): res = the block output, res2 = the BatchNorm input of the residual tail; pool: that epilogue also writes the
    2x2 sum-pool of its result into this [n, h/2, w/2, cout] tensor (lib.ctl_conv_pool_ok(desc) says whether the problem can);
    xout (with x2): the virtual input the conv stages is also written to this tensor of x's shape and dtype."""
    require_gpu(x, wpack)
    if (x2 is not None) != (int(d["pro_affine"]) == 2) or (x2 is not None and (x2.dtype != x.dtype or x2.shape != x.shape)):
        raise _ffi.CtlError("conv_forward: x2 goes with ctl_conv.pro_affine == 2 and has the dtype and shape of x")
    n, cout, oh, ow = int(d["n"]), int(d["cout"]), int(d["out_h"]), int(d["out_w"])
    dt = int(d["dt"])
    for t, flag, what in ((x, _ffi.DT_X16, "x"), (res, _ffi.DT_RES16, "res"), (res2, _ffi.DT_RES16, "res2"), (y, _ffi.DT_Y16, "y")):
        if t is not None and (t.dtype == torch.bfloat16) != bool(dt & flag):
            raise _ffi.CtlError(f"conv_forward: dtype of {what} ({t.dtype}) disagrees with ctl_conv.dt = {dt}")
    if y is None:
        y = empty_nhwc(n, cout, oh, ow, x.device, torch.bfloat16 if dt & _ffi.DT_Y16 else torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty(lib.ctl_conv_stats_floats(_ffi.desc_ptr(d)), dtype=torch.float32, device=x.device)
    check(lib.ctl_conv_forward_ex(_ffi.desc_ptr(d), ptr(x), ptr(wpack), ptr(bias), ptr(pro_scale), ptr(pro_shift), ptr(res),
                                  ptr(res_scale), ptr(res_shift), ptr(res2), ptr(x2), ptr(y), ptr(stats), ptr(pool), ptr(xout), stream_ptr()), "ctl_conv_forward")
    return y, stats


def conv_wgrad(d: np.ndarray, x, dy, dw: torch.Tensor, strides, dbias: Optional[torch.Tensor] = None, pro_scale=None,
               pro_shift=None, accumulate=False, dy2=None, dy_coef=None):
    """dy2 / dy_coef: the output gradient is the virtual BatchNorm-backward result A*dy + B*dy2 + C (ctl_conv_wgrad_ex)."""
    require_gpu(x, dy, dw)
    if dy2 is not None and (dy_coef is None or dy2.dtype != dy.dtype or dy2.shape != dy.shape):
        raise _ffi.CtlError("conv_wgrad: dy2 needs dy_coef and the dtype and shape of dy")
    dt = int(d["dt"])
    if (x.dtype == torch.bfloat16) != bool(dt & _ffi.DT_X16) or (dy.dtype == torch.bfloat16) != bool(dt & _ffi.DT_Y16):
        raise _ffi.CtlError(f"conv_wgrad: dtypes of x / dy ({x.dtype}, {dy.dtype}) disagree with ctl_conv.dt = {dt}")
    dp = _ffi.desc_ptr(d)
    wpart = torch.empty(lib.ctl_wgrad_partial_floats(dp), dtype=torch.float32, device=x.device)
    bpart = torch.empty(lib.ctl_wgrad_bias_partial_floats(dp), dtype=torch.float32, device=x.device) if dbias is not None else None
    check(lib.ctl_conv_wgrad_ex(dp, ptr(x), ptr(pro_scale), ptr(pro_shift), ptr(dy), ptr(dy2), ptr(dy_coef), ptr(wpart), ptr(bpart),
                                stream_ptr()), "ctl_conv_wgrad")
    check(lib.ctl_wgrad_reduce(dp, ptr(wpart), ptr(bpart), ptr(dw), *[int(s) for s in strides], ptr(dbias), int(accumulate),
                               stream_ptr()), "ctl_wgrad_reduce")
    return dw, dbias


# ---------------------------------------------------------------------------------------------- BN pieces
def bn_finalize(partial, c, count, gamma, beta, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, nbt=None, groups=1):
    """partial: [groups][blocks][2][c]; `count` = pixels of one group.  Returns scale, shift, mean, invstd, each [groups*c]."""
    dev = partial.device
    scale, shift, mean, invstd = (torch.empty(groups * c, dtype=torch.float32, device=dev) for _ in range(4))
    blocks = partial.numel() // (2 * c * groups)
    check(lib.ctl_bn_finalize(ptr(partial), blocks, c, count, ptr(gamma), ptr(beta), eps, momentum,
                              int(running_mean is not None), ptr(running_mean), ptr(running_var), ptr(nbt), ptr(scale),
                              ptr(shift), ptr(mean), ptr(invstd), groups, stream_ptr()), "ctl_bn_finalize")
    return scale, shift, mean, invstd


def bn_act(x, scale, shift, slope, groups=1):
    y = torch.empty_like(x)
    n, c, h, w = x.shape
    check(lib.ctl_bn_act(ptr(x), ptr(scale), ptr(shift), slope, ptr(y), n * h * w, c, groups, stream_ptr()), "ctl_bn_act")
    return y


# ---------------------------------------------------------------------------------------------- STN input / losses
def softmax_t_fwd(x: torch.Tensor, temperature: float = 2.0) -> torch.Tensor:
    require_gpu(x)
    n, c, h, w = x.shape
    p = torch.empty_like(x)
    check(lib.ctl_softmax_t_fwd(ptr(x), 1.0 / temperature, ptr(p), n * h * w, c, stream_ptr()), "ctl_softmax_t_fwd")
    return p


def softmax_t_bwd(p, dp, temperature: float = 2.0):
    n, c, h, w = p.shape
    dx = torch.empty_like(p)
    check(lib.ctl_softmax_t_bwd(ptr(p), ptr(dp), 1.0 / temperature, ptr(dx), n * h * w, c, stream_ptr()), "ctl_softmax_t_bwd")
    return dx


def onehot(label: torch.Tensor, c: int) -> torch.Tensor:
    require_gpu(label)
    label = label.long().contiguous()
    n, h, w = label.shape
    y = empty_nhwc(n, c, h, w, label.device)
    check(lib.ctl_onehot(ptr(label), ptr(y), n * h * w, c, stream_ptr()), "ctl_onehot")
    return y


def ce2d_fwd(logit, label):
    n, c, h, w = logit.shape
    partial = torch.empty(_ffi.RED_BLOCKS, dtype=torch.float64, device=logit.device)
    loss = torch.empty((), dtype=torch.float32, device=logit.device)
    check(lib.ctl_ce2d_fwd(ptr(logit), ptr(label), n * h * w, c, ptr(partial), ptr(loss), stream_ptr()), "ctl_ce2d_fwd")
    return loss


def ce2d_bwd(logit, label, gout):
    n, c, h, w = logit.shape
    d = torch.empty_like(logit)
    check(lib.ctl_ce2d_bwd(ptr(logit), ptr(label), ptr(gout), n * h * w, c, ptr(d), stream_ptr()), "ctl_ce2d_bwd")
    return d


def mse_fwd(a, b, scale):
    partial = torch.empty(_ffi.RED_BLOCKS, dtype=torch.float64, device=a.device)
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    check(lib.ctl_mse_fwd(ptr(a), ptr(b), a.numel(), scale, ptr(partial), ptr(loss), stream_ptr()), "ctl_mse_fwd")
    return loss


def mse_bwd(a, b, gout, scale):
    d = torch.empty_like(a)
    check(lib.ctl_mse_bwd(ptr(a), ptr(b), ptr(gout), a.numel(), scale, ptr(d), stream_ptr()), "ctl_mse_bwd")
    return d


def argmax_c(logit: torch.Tensor) -> torch.Tensor:
    require_gpu(logit)
    logit = as_nhwc(logit)
    n, c, h, w = logit.shape
    out = torch.empty((n, h, w), dtype=torch.uint8, device=logit.device)
    check(lib.ctl_argmax_c(ptr(logit), ptr(out), n * h * w, c, stream_ptr()), "ctl_argmax_c")
    return out


# ---------------------------------------------------------------------------------------------- latent masking
def latent_score(grad: torch.Tensor, mode: int) -> torch.Tensor:
    """mode 0: [N,C] signed mean over H*W; mode 1: [N,H*W] signed mean over C (model_util.py:224-225 / 285-286)."""
    require_gpu(grad)
    grad = as_nhwc(grad)
    n, c, h, w = grad.shape
    L = c if mode == 0 else h * w
    score = torch.empty((n, L), dtype=torch.float32, device=grad.device)
    ws = lib.ctl_latent_score_ws_floats(mode, n, h * w, c)
    scratch = torch.empty(max(ws, 1), dtype=torch.float32, device=grad.device)
    check(lib.ctl_latent_score(mode, ptr(grad), ptr(score), ptr(scratch), n, h * w, c, stream_ptr()), "ctl_latent_score")
    return score


def latent_mask_apply(code: torch.Tensor, score: torch.Tensor, mode: int, k, soft_noise: Optional[torch.Tensor] = None):
    """Returns (masked code [N,C,H,W], mask [N,C,1,1] or [N,1,H,W]).  `k` is an int or a 1-element int32 device tensor."""
    require_gpu(code, score)
    code = as_nhwc(code)
    n, c, h, w = code.shape
    masked = torch.empty_like(code)
    L = c if mode == 0 else h * w
    mask = torch.empty((n, L), dtype=torch.float32, device=code.device)
    k_dev = k if isinstance(k, torch.Tensor) else None
    k_host = 0 if k_dev is not None else int(k)
    if soft_noise is not None:
        soft_noise = soft_noise.reshape(n, L).float().contiguous()
    ws = lib.ctl_latent_mask_apply_ws_floats(mode, n, h * w, c)
    scratch = torch.empty(ws, dtype=torch.float32, device=code.device) if ws else None
    check(lib.ctl_latent_mask_apply(mode, ptr(code), ptr(score), ptr(soft_noise), k_host, ptr(k_dev), ptr(masked), ptr(mask),
                                    ptr(scratch), n, h * w, c, stream_ptr()), "ctl_latent_mask_apply")
    return masked, (mask.view(n, c, 1, 1) if mode == 0 else mask.view(n, 1, h, w))


def latent_mask(grad: torch.Tensor, code: torch.Tensor, mode: int, k, soft_noise: Optional[torch.Tensor] = None, want_score: bool = False):
    """The generator tail behind one call (`ctl_latent_mask_fused`): signed-mean score of `grad`, rank-select of the top-k entries
    (strict '>', exact under ties), masked = code * mask; ONE launch at latent-code sizes.  Returns (masked, mask[, score]); mask is
    [N,C,1,1] (mode 0) or [N,1,H,W] (mode 1).  `k`: int or 1-element int32 device tensor (graph replay)."""
    require_gpu(grad, code)
    grad, code = as_nhwc(grad), as_nhwc(code)
    n, c, h, w = code.shape
    if grad.shape != code.shape:
        raise ValueError("latent_mask: grad and code must have the same shape")
    L = c if mode == 0 else h * w
    masked = torch.empty_like(code)
    mask = torch.empty((n, L), dtype=torch.float32, device=code.device)
    score = torch.empty((n, L), dtype=torch.float32, device=code.device) if want_score else None
    k_dev = k if isinstance(k, torch.Tensor) else None
    k_host = 0 if k_dev is not None else int(k)
    if soft_noise is not None:
        soft_noise = soft_noise.reshape(n, L).float().contiguous()
    need = lib.ctl_latent_mask_fused_ws_floats(mode, n, h * w, c)
    ws = torch.empty(need, dtype=torch.float32, device=code.device) if need else None
    check(lib.ctl_latent_mask_fused(mode, ptr(grad), ptr(code), ptr(soft_noise), k_host, ptr(k_dev), ptr(masked), ptr(mask), ptr(score),
                                    ptr(ws), n, h * w, c, stream_ptr()), "ctl_latent_mask_fused")
    mask = mask.view(n, c, 1, 1) if mode == 0 else mask.view(n, 1, h, w)
    return (masked, mask, score) if want_score else (masked, mask)


def dropout2d(z: torch.Tensor, p: float, keep: Optional[torch.Tensor] = None, seed: int = 0, state: Optional[torch.Tensor] = None,
              want_mask: bool = False):
    """Returns (out, keep[N,C]) or, with want_mask, (out, keep, mask) where mask is upstream's full-size equality mask
    (model.py:334-336).  keep=None draws the Bernoulli pattern on device from `seed`; with `state` (device int64[3], see
    ctl_step_tick) `seed` is a call-site salt and nothing step-dependent is baked into the launch (HIP-graph replay)."""
    require_gpu(z)
    z = as_nhwc(z)
    n, c, h, w = z.shape
    out = torch.empty_like(z)
    keep_out = torch.empty((n, c), dtype=torch.float32, device=z.device)
    mask = torch.empty_like(z) if want_mask else None
    if keep is not None:
        keep = keep.reshape(n, c).float().contiguous()
    check(lib.ctl_dropout2d_ex(ptr(z), ptr(keep), seed & (2 ** 64 - 1), ptr(state), p, ptr(out), ptr(keep_out), ptr(mask), n, h * w, c,
                               stream_ptr()), "ctl_dropout2d_ex")
    return (out, keep_out, mask) if want_mask else (out, keep_out)


def uniform(shape, device, seed: int, state: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device)
    if state is not None:
        check(lib.ctl_uniform_dev(ptr(out), out.numel(), seed & (2 ** 64 - 1), ptr(state), stream_ptr()), "ctl_uniform_dev")
    else:
        check(lib.ctl_uniform(ptr(out), out.numel(), seed & (2 ** 64 - 1), stream_ptr()), "ctl_uniform")
    return out


def step_tick(state: torch.Tensor) -> None:
    """Advance the device-resident step state (RNG counter, Adam step): one launch at the head of a graph-replayed step."""
    require_gpu(state)
    assert state.dtype == torch.int64 and state.numel() >= 3
    check(lib.ctl_step_tick(ptr(state), stream_ptr()), "ctl_step_tick")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, state: Optional[torch.Tensor] = None):
    require_gpu(p, g, m, v)
    if state is not None:       # step count read from state[2] on the device
        check(lib.ctl_adam_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, ptr(state), grad_scale, stream_ptr()),
              "ctl_adam_dev")
    else:
        check(lib.ctl_adam(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, stream_ptr()),
              "ctl_adam")


# ---------------------------------------------------------------------------------------------- metrics + input pipeline
def confusion_hist(label_true: torch.Tensor, label_pred: torch.Tensor, n_class: int, hist: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Accumulate `runningScore._fast_hist` (metrics.py:18-23) into `hist` (int64 [n_class, n_class], created zeroed if None)."""
    require_gpu(label_true, label_pred)
    lt = label_true.long().contiguous()
    lp = label_pred.to(torch.uint8).contiguous()
    if lt.numel() != lp.numel():
        raise ValueError("confusion_hist: label_true and label_pred differ in size")
    if hist is None:
        hist = torch.zeros((n_class, n_class), dtype=torch.int64, device=lt.device)
    check(lib.ctl_confusion_hist(ptr(lt), ptr(lp), lt.numel(), n_class, ptr(hist), stream_ptr()), "ctl_confusion_hist")
    return hist


def rescale_intensity(data: torch.Tensor, new_min: float = 0.0, new_max: float = 1.0, eps: float = 1e-20) -> torch.Tensor:
    """basic_operations.py:232-245 on device; data: [N,C,H,W] float32 in plain NCHW memory (each (n,c) plane contiguous)."""
    require_gpu(data)
    x = data.float().contiguous()
    n, c, h, w = x.shape
    out = torch.empty_like(x)
    ws = torch.empty(lib.ctl_rescale_intensity_ws_floats(n * c), dtype=torch.float32, device=x.device)
    check(lib.ctl_rescale_intensity(ptr(x), ptr(out), ptr(ws), n * c, h * w, new_min, new_max, eps, stream_ptr()), "ctl_rescale_intensity")
    return out


def noise_clamp(x: torch.Tensor, noise: Optional[torch.Tensor] = None, sigma: float = 0.05, lo: float = 0.0, hi: float = 1.0,
                seed: int = 0) -> torch.Tensor:
    """clamp(x + noise, lo, hi); noise=None draws sigma*N(0,1) on device from `seed` (train...py:185-187)."""
    require_gpu(x)
    x = x.float().contiguous()
    if noise is not None:
        noise = noise.float().contiguous()
        if noise.shape != x.shape:
            raise ValueError("noise_clamp: noise must have the shape of x")
    out = torch.empty_like(x)
    check(lib.ctl_noise_clamp(ptr(x), ptr(noise), seed & (2 ** 64 - 1), sigma, lo, hi, ptr(out), x.numel(), stream_ptr()), "ctl_noise_clamp")
    return out


def crop_or_pad(image: torch.Tensor, crop_size, label: Optional[torch.Tensor] = None):
    """basic_operations.py:173-220 for [n,h,w] (or [h,w]) device tensors: (image, label) cropped / zero-padded to crop_size."""
    def one(a):
        require_gpu(a)
        squeeze = a.dim() == 2
        a3 = (a.unsqueeze(0) if squeeze else a).contiguous()
        if a3.dim() != 3 or a3.element_size() not in (1, 4, 8):
            raise ValueError("crop_or_pad: expected a [n,h,w] or [h,w] tensor with 1-, 4- or 8-byte elements")
        n, h, w = a3.shape
        out = torch.empty((n, int(crop_size[0]), int(crop_size[1])), dtype=a3.dtype, device=a3.device)
        check(lib.ctl_crop_or_pad(ptr(a3), ptr(out), a3.element_size(), n, h, w, int(crop_size[0]), int(crop_size[1]), stream_ptr()),
              "ctl_crop_or_pad")
        return out[0] if squeeze else out
    return one(image), (None if label is None else one(label))
