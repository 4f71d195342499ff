"""torch.autograd bridge: one Function per *network pass* (not per layer) plus the fused loss / STN-input ops, so the
reference's `loss.backward()` call pattern works unchanged while every FLOP runs in the HIP kernels."""
from __future__ import annotations

import torch

from . import ops
from ._ffi import CtlError


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.float().contiguous(memory_format=torch.channels_last)


class _NetFn(torch.autograd.Function):
    """A whole encoder / decoder pass.  Inputs: the activation `x` and the network's flat parameter buffer."""

    @staticmethod
    def forward(ctx, net, mode, affine, groups, x, flat):
        outs, act, plan = net.run_forward(x, mode, groups)
        net.remember_pass(x, act, outs, plan, mode, groups)
        net._pass_seq += 1
        ctx.counted = bool(flat.requires_grad)            # this pass owes the network a parameter gradient
        if ctx.counted:
            net._pending_bwd += 1
        ctx.net, ctx.mode, ctx.affine, ctx.act, ctx.plan, ctx.seq = net, mode, affine, act, plan, net._pass_seq
        ctx.save_for_backward(x, *outs)
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        if ctx.mode == "C":
            raise CtlError("backward through an eval-mode (running-statistics) pass is not part of the hot path")
        saved = ctx.saved_tensors
        x, outs = saved[0], saved[1:]
        need_dx, need_w = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        if all(d is None for d in douts) or not (need_dx or need_w):
            return None, None, None, None, None, None
        douts = tuple(None if d is None else _nhwc(d) for d in douts)
        dx, gflat = ctx.net.run_backward(x, ctx.act, outs, ctx.plan, ctx.mode, douts, need_dx, need_w, ctx.affine)
        if gflat is not None and ctx.net._defer_grads:
            # parked for ONE accumulation per network after backward (CtlNet.collect_deferred_grads): handing it to autograd would make
            # every pass a cross-stream dependency on the leaf's accumulation stream when the step runs as two launch chains
            ev = torch.cuda.Event()
            ev.record()
            ctx.net._deferred.append((ctx.seq, gflat, ev))
            gflat = None
        if ctx.counted and need_w:
            net = ctx.net
            net._pending_bwd -= 1
            if net._pending_bwd == 0 and net._on_grads_complete is not None and net._defer_grads and \
                    (net._stack is None or net._stack.done or not net._stack.filled):
                # the last backward pass of this network in this step: its gradient range is complete once the parked per-pass
                # gradients are added; dist.DataParallel does that here, on this pass' stream, and starts the exchange of the range behind it
                net._on_grads_complete()
        return None, None, None, None, dx, gflat


class _SlotFn(torch.autograd.Function):
    """A network pass that is one slot of a PassStack (nets.py): the forward runs now, into the stacked arena; the backward node only
    RECORDS the gradients arriving at the pass' outputs.  The solver issues the stacked backward of all slots once every slot has its
    gradients (CtlNet.backward_stack) and hands the input gradients back to autograd from the pass' input tensors."""

    @staticmethod
    def forward(ctx, net, stack, mode, groups, x, flat):
        p = stack.filled
        outs, act, plan = net.run_forward(x, mode, groups, stack=stack)
        net.remember_pass(x, act, outs, plan, mode, groups)
        net._pass_seq += 1
        stack.seqs.append(net._pass_seq)
        stack.need_dx.append(bool(ctx.needs_input_grad[4]))
        ctx.stack, ctx.p = stack, p
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        if all(d is None for d in douts):      # (a sweep that reaches this node without a gradient for it: nothing to record)
            return None, None, None, None, None, None
        if ctx.stack.done:
            raise CtlError("a stacked pass received a gradient after its backward had been issued")
        ctx.stack.receive(ctx.p, douts)
        return None, None, None, None, None, None


def net_apply(net, x: torch.Tensor, groups: int = 1):
    """Run `net` on `x` (logical NCHW) in its current BatchNorm mode; returns a tuple of outputs.  groups > 1: `x` stacks that
    many independent batches along n and BatchNorm handles each as its own call (one launch chain instead of `groups`)."""
    ops.require_gpu(x)
    x = _nhwc(x)
    mode = net.bn_mode()
    track_params = torch.is_grad_enabled() and mode != "C" and net.wants_param_grad()
    flat = net._flat if track_params else net._flat.detach()
    stack = net._stack
    if stack is not None and track_params and stack.accepts(x.shape[0], x.shape[2], x.shape[3], groups):
        outs = _SlotFn.apply(net, stack, mode, groups, x, flat)
        stack.roots.append(x)
        return outs
    return _NetFn.apply(net, mode, mode == "A", groups, x, flat)


class _SplitHalves(torch.autograd.Function):
    """(x[:n], x[n:]) of a stacked (grouped) pass.  Plain slicing would do, but its backward builds two zero-filled NCHW tensors of the
    full stacked size, copies a half into each, adds them and leaves the sum for a layout conversion: ~100 us of fills and copies
    per step.  Here the incoming halves are copied straight into one NHWC tensor."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        ctx.meta = (x.shape, x.device)
        n = x.shape[0] // 2
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None
        shape, device = ctx.meta
        n = shape[0] // 2
        g = torch.empty(shape, dtype=torch.float32, device=device, memory_format=torch.channels_last)
        for half, gh in ((g[:n], ga), (g[n:], gb)):
            if gh is None:
                half.zero_()
            else:
                half.copy_(gh)
        return g


def split_halves(x: torch.Tensor):
    return _SplitHalves.apply(x)


class _CrossEntropy2D(torch.autograd.Function):
    """cross_entropy_2D with an integer label map (custom_loss.py:706-740 / model_util.py:104-115)."""

    @staticmethod
    def forward(ctx, logit, label):
        logit = _nhwc(logit)
        label = label.long().contiguous()
        ctx.save_for_backward(logit, label)
        return ops.ce2d_fwd(logit, label)

    @staticmethod
    def backward(ctx, g):
        logit, label = ctx.saved_tensors
        return ops.ce2d_bwd(logit, label, g.float().contiguous()), None


class _ScaledMSE(torch.autograd.Function):
    """scale * mean((a-b)^2)"""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = _nhwc(a), _nhwc(b)
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        return ops.mse_fwd(a, b, scale)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ops.mse_bwd(a, b, g.float().contiguous(), ctx.scale), None, None


class _SoftmaxT(torch.autograd.Function):
    """softmax(x / T, dim=C) (construct_input, basic_operations.py:129-131)."""

    @staticmethod
    def forward(ctx, x, temperature):
        p = ops.softmax_t_fwd(_nhwc(x), temperature)
        ctx.save_for_backward(p)
        ctx.t = temperature
        return p

    @staticmethod
    def backward(ctx, dp):
        p, = ctx.saved_tensors
        return ops.softmax_t_bwd(p, _nhwc(dp), ctx.t), None


def cross_entropy_2D(logit, label):
    ops.require_gpu(logit, label)
    return _CrossEntropy2D.apply(logit, label)


def scaled_mse(a, b, scale=1.0):
    ops.require_gpu(a, b)
    return _ScaledMSE.apply(a, b, float(scale))


def softmax_t(x, temperature=2.0):
    ops.require_gpu(x)
    return _SoftmaxT.apply(x, float(temperature))
