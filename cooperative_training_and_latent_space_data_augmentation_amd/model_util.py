"""Latent-space masking API of the reference (medseg/models/model_util.py), backed by the HIP kernels.

Same names and argument meaning as upstream: `mask_latent_code_channel_wise` (:180-255), `mask_latent_code_spatial_wise`
(:258-318), `_disable_tracking_bn_stats` (:414-451), `set_grad` (:163-165), `makeVariable` (:603-618),
`make_one_hot` (:168-177), `cross_entropy_2D` (:104-115).  Extra keyword-only arguments `k` / `soft_noise` inject the
draws that upstream takes from numpy / torch RNGs (parity tests; CUDA-graph style replays).
"""
from __future__ import annotations

import contextlib
from typing import Optional

import numpy as np
import torch

from . import ops
from .autograd import cross_entropy_2D as _ce2d, scaled_mse as _mse


def set_grad(module, requires_grad=False):
    plist = module.param_list() if hasattr(module, "param_list") else module.parameters()      # (cached list of the engine's networks)
    for p in plist:
        p.requires_grad = requires_grad


def makeVariable(tensor, use_gpu=True, type="long", requires_grad=True):
    t = tensor.data
    if type == "long":
        t = t.long()
    elif type == "float":
        t = t.float()
    else:
        raise NotImplementedError
    if use_gpu:
        t = t.cuda()
    return t.detach().requires_grad_(requires_grad) if t.is_floating_point() else t.detach()


def make_one_hot(y, num_classes=4):
    return ops.onehot(y, num_classes)


def cross_entropy_2D(input, target, weight=None, size_average=True):
    if weight is not None or not size_average or target.dim() != 3:
        raise NotImplementedError("only the un-weighted, averaged, label-map form is on the hot path")
    return _ce2d(input, target)


@contextlib.contextmanager
def _disable_tracking_bn_stats(model):
    """BatchNorm 'mode B': batch statistics, no running-stat update, gamma/beta frozen for this pass."""
    old = model._bn_track
    bn_params = model.__dict__.get("_bn_plist")           # (cached: the module tree is fixed; the walk cost 0.3 ms of host time per step)
    if bn_params is None:
        bn_params = model.__dict__["_bn_plist"] = [p for m in model.modules() if type(m).__name__ == "_BNP" for p in (m.weight, m.bias)]
    model._bn_track = False
    for p in bn_params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        model._bn_track = old
        for p in bn_params:       # upstream restores requires_grad to the saved *track* flag
            p.requires_grad_(bool(old))


def _draw_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())        # host RNG: respects torch.manual_seed, no device sync


REUSE_SALIENCY_FORWARD = True     # see _saliency_grad


def _saliency_grad(code, decoder_function, label, num_classes, loss_type):
    """dL/dz through the (frozen) decoder: forward + dgrad-only backward (model_util.py:202-223).

    Upstream's saliency pass runs `decoder_function(code)` in training mode on the very tensor the standard pass has just decoded
    (model.py:436-447 store z_i / z_s, model.py:491-501 hands them to perturb_latent_code): same input, same weights, same mode -- the same
    activations, the same batch statistics.  If the decoder still remembers that pass (CtlNet.reuse_pass) its activations are re-used:
    the loss gradient w.r.t. the output is formed from the remembered output, the data-gradient-only backward runs on the remembered
    activations, and the second running-statistics update of every BatchNorm is replayed from the saved batch statistics -- bit for
    bit the numbers of running the pass again (tests/test_engine_gpu.py::test_saliency_forward_reuse_is_bitwise), one decoder forward less."""
    reuse = decoder_function.reuse_pass(ops.as_nhwc(code.detach())) if (REUSE_SALIENCY_FORWARD and hasattr(decoder_function, "reuse_pass")) else None
    if reuse is not None and loss_type in ("ce", "mse", "corr"):
        outs, backward, handle = reuse
        out = outs[0]
        one = torch.ones((), device=out.device)
        if loss_type == "ce":
            dout = ops.ce2d_bwd(out, label.long().contiguous(), one)
        else:
            gt = ops.as_nhwc(ops.onehot(label, num_classes) if label.dim() < code.dim() else label)
            dout = ops.mse_bwd(out, gt, one, 1.0) if loss_type == "mse" else gt / float(out.numel())
        grad = backward((dout,))
        decoder_function.replay_running_stats(handle)
        return ops.as_nhwc(code.detach()), grad
    code = ops.as_nhwc(code.detach()).requires_grad_(True)
    with torch.enable_grad():
        out = decoder_function(code)
        if loss_type == "ce":
            loss = _ce2d(out, label)
            grad = torch.autograd.grad(loss, [code])[0]
        else:
            gt = ops.onehot(label, num_classes) if label.dim() < code.dim() else label
            if loss_type == "mse":
                loss = _mse(out, gt, 1.0)
                grad = torch.autograd.grad(loss, [code])[0]
            elif loss_type == "corr":     # d mean(out*gt) / d out = gt / numel
                grad = torch.autograd.grad(out, [code], grad_outputs=ops.as_nhwc(gt) / float(out.numel()))[0]
            else:
                raise NotImplementedError(loss_type)
    return code.detach(), grad


def _mask(latent_code, decoder_function, label, num_classes, percentile, random, loss_type, if_detach, if_soft, mode,
          k, soft_noise):
    ops.require_gpu(latent_code)
    code, grad = _saliency_grad(latent_code, decoder_function, label, num_classes, loss_type)
    n, c, h, w = code.shape
    L = c if mode == 0 else h * w
    if k is None:
        if random:
            percentile = np.random.rand() * percentile
        k = int(L * percentile)
    if if_soft and soft_noise is None:
        soft_noise = ops.uniform((n, L), code.device, _draw_seed())
    # score -> rank-select -> apply in ONE launch (the three-launch path ops.latent_score + ops.latent_mask_apply computes the same bits)
    masked, mask = ops.latent_mask(grad, code, mode, k, soft_noise if if_soft else None)
    if not if_detach:
        masked = latent_code * mask
    if hasattr(decoder_function, "zero_grad"):
        try:
            decoder_function.zero_grad()
        except Exception:
            pass
    return masked, mask


def mask_latent_code_channel_wise(latent_code, decoder_function, label, num_classes=2, percentile=1 / 3.0, random=False,
                                  loss_type="corr", if_detach=True, if_soft=False, *, k: Optional[int] = None,
                                  soft_noise: Optional[torch.Tensor] = None):
    """Mask the top-k channels ranked by the signed mean of dL/dz over H*W; mask [N,C,1,1]."""
    return _mask(latent_code, decoder_function, label, num_classes, percentile, random, loss_type, if_detach, if_soft, 0, k,
                 soft_noise)


def mask_latent_code_spatial_wise(latent_code, decoder_function, label, num_classes, percentile=1 / 3.0, random=False,
                                  loss_type="corr", if_detach=True, if_soft=False, *, k: Optional[int] = None,
                                  soft_noise: Optional[torch.Tensor] = None):
    """Mask the top-k positions ranked by the signed mean of dL/dz over C; mask [N,1,H,W]."""
    return _mask(latent_code, decoder_function, label, num_classes, percentile, random, loss_type, if_detach, if_soft, 1, k,
                 soft_noise)
