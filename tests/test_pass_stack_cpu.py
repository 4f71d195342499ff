"""Round 6, nets.PassStack (CPU, plan structure only): the passes of one network in one step write their activations into the slots of ONE
arena laid out as if a single pass had run on the stacked batch, so that the backward of all passes is one launch chain.

Checked here without a GPU: the slot plans' layout IS the layout of the plan compiled directly for the stacked batch (same arena size, same
tensor and coefficient-table offsets), pass p sits p * (tensor size) behind pass 0, the stacked backward plan is op for op the backward plan of
the directly compiled stacked forward, and a pass in BatchNorm mode B clears its groups' bits in the gamma / beta gradient mask."""
import numpy as np
import pytest
import torch

from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, nets
from cooperative_training_and_latent_space_data_augmentation_amd.nets import T, _StackedForward, _stack_rec


def compile_forward(net, n, h, w, mode, groups=1, pp=(0, 1)):
    net._cur_groups, net._cur_pp = groups, pp
    try:
        return net._compile_forward(n, h, w, mode)
    finally:
        net._cur_groups, net._cur_pp = 1, (0, 1)


def compile_backward(net, fwd, groups, mask, need_dx, amask=0):
    net._cur_groups, net._cur_affine_mask = groups, amask
    try:
        return net._compile_backward(fwd, "A", mask, need_dx, True, True)
    finally:
        net._cur_groups, net._cur_affine_mask = 1, 0


def tensors_of(obj, out):
    if isinstance(obj, T):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            tensors_of(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            tensors_of(v, out)
    return out


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name,groups", [("image_encoder", 1), ("segmentation_decoder", 1), ("image_decoder", 1), ("shape_encoder", 2), ("shape_decoder", 2)])
def test_slot_plans_share_the_layout_of_the_stacked_plan(name, groups, dtype):
    net = nets.build_networks(device="cpu", dtype=dtype)[name]
    n = 4
    h = w = 64 if name.endswith("encoder") else 4
    p0 = compile_forward(net, n, h, w, "A", groups, (0, 2))
    p1 = compile_forward(net, n, h, w, "B", groups, (1, 2))
    direct = compile_forward(net, 2 * n, h, w, "A", 2 * groups)
    assert p0.act_bytes == p1.act_bytes == direct.act_bytes
    assert _stack_rec(p0.rec, 2) == direct.rec, "stacked view of slot 0 == the record of the directly compiled stacked pass"
    t0, t1 = tensors_of(p0.rec, []), tensors_of(p1.rec, [])
    assert len(t0) == len(t1) > 10
    for a, b in zip(t0, t1):
        assert (a.n, a.h, a.w, a.c, a.b16, a.ref[0]) == (b.n, b.h, b.w, b.c, b.b16, b.ref[0])
        if a.ref[0] == nets.S_ACT:
            assert b.ref[1] - a.ref[1] == (2 if a.b16 else 4) * a.n * a.h * a.w * a.c, "pass 1 sits one pass' images behind pass 0"
        else:
            assert b.ref == a.ref                      # external tensors (input / outputs): the caller hands in the slot's view
    # the coefficient tables: pass 1's rows follow pass 0's `groups` rows
    for (bn0, co0), (bn1, co1) in zip(p0.bn_log, p1.bn_log or p0.bn_log):
        assert bn0 is bn1
    co_a = [v for k, v in p0.rec.items() if k.startswith("co")] if "co0" in p0.rec else [p0.rec["blocks"][0]["co1"]]
    co_b = [v for k, v in p1.rec.items() if k.startswith("co")] if "co0" in p1.rec else [p1.rec["blocks"][0]["co1"]]
    for a, b in zip(co_a, co_b):
        for key in ("scale", "shift", "mean", "invstd", "uvar"):
            assert b[key][0] == a[key][0] == nets.S_ACT and b[key][1] > a[key][1]
    # backward: the stacked view compiles to the plan of the direct stacked pass, record for record
    mask = (True, True) if name == "image_encoder" else (True,)
    need_dx = not name == "image_encoder"
    b_view = compile_backward(net, _StackedForward(_stack_rec(p0.rec, 2), 2 * groups), 2 * groups, mask, need_dx)
    b_direct = compile_backward(net, direct, 2 * groups, mask, need_dx)
    assert b_view.n_ops == b_direct.n_ops and b_view.ops.tobytes() == b_direct.ops.tobytes()
    assert b_view.bscr_bytes == b_direct.bscr_bytes
    # one launch chain instead of two: the stacked plan has the op count of ONE per-pass plan
    # (four groups of 128 channels pass the kernels' 256-entry coefficient tables: those layers fall back to stored apply passes)
    if groups == 1:
        b_single = compile_backward(net, p0, groups, mask, need_dx)
        assert b_view.n_ops <= b_single.n_ops + 2


def test_mode_b_passes_are_masked_out_of_the_affine_gradients():
    net = nets.build_networks(device="cpu")["image_encoder"]
    p0 = compile_forward(net, 2, 32, 32, "A", 1, (0, 2))
    view = _StackedForward(_stack_rec(p0.rec, 2), 2)
    b = compile_backward(net, view, 2, (True, True), False, amask=0b01)
    fin = [o for o in b.ops if int(o["kind"]) == _ffi.OP_BN_BWD_FINALIZE]
    assert len(fin) >= 10
    for o in fin:
        assert int(o["i"][2]) == 2 and int(o["i"][4]) == 0b01 and int(o["slot"][5]) >= 0      # two groups, group 0 only adds to dgamma / dbeta
    b_all = compile_backward(net, view, 2, (True, True), False, amask=0)
    assert all(int(o["i"][4]) == 0 for o in b_all.ops if int(o["kind"]) == _ffi.OP_BN_BWD_FINALIZE)


def test_gather_stacked_uses_views_of_one_allocation_without_a_copy():
    alloc = lambda s: torch.empty((s[0], s[3], s[1], s[2]), memory_format=torch.channels_last)
    base = torch.randn(8, 16, 4, 4).contiguous(memory_format=torch.channels_last)
    a, b = base[0:4], base[4:8]
    g = nets.CtlNet._gather_stacked([a, b], alloc)
    assert g.data_ptr() == base.data_ptr() and g.shape == base.shape
    c = torch.randn(4, 16, 4, 4).contiguous(memory_format=torch.channels_last)
    g2 = nets.CtlNet._gather_stacked([a, c], alloc)
    assert g2.data_ptr() not in (base.data_ptr(), c.data_ptr()) and torch.equal(g2[:4], a) and torch.equal(g2[4:], c)
    g3 = nets.CtlNet._gather_stacked([None, c], alloc)
    assert torch.equal(g3[:4], torch.zeros_like(c)) and torch.equal(g3[4:], c)
    # the second half first: not the stacked order -> copied
    g4 = nets.CtlNet._gather_stacked([b, a], alloc)
    assert torch.equal(g4[:4], b) and torch.equal(g4[4:], a) and g4.data_ptr() != base.data_ptr()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["image_encoder", "segmentation_decoder", "image_decoder"])
def test_split_backward_plan_is_the_plan_with_the_weight_gradient_family_moved_behind(name, dtype):
    """Plan.main_ops | Plan.tail_ops: every record of the plan exactly once, both parts in plan order; the tail holds the weight-gradient
    family (and the transposed convs' bias sums with their reduction) and nothing the data-gradient chain needs."""
    net = nets.build_networks(device="cpu", dtype=dtype)[name]
    h = w = 64 if name.endswith("encoder") else 4
    p0 = compile_forward(net, 4, h, w, "A", 1, (0, 2))
    view = _StackedForward(_stack_rec(p0.rec, 2), 2)
    mask = (True, True) if name == "image_encoder" else (True,)
    net._cur_split = True
    try:
        b = compile_backward(net, view, 2, mask, name != "image_encoder")
    finally:
        net._cur_split = False
    assert b.main_ops is not None and b.tail_ops is not None and len(b.main_ops) + len(b.tail_ops) == b.n_ops
    from collections import Counter
    key = lambda o: o.tobytes()
    assert Counter(key(o) for o in b.ops) == Counter(key(o) for o in b.main_ops) + Counter(key(o) for o in b.tail_ops)
    for part in (b.main_ops, b.tail_ops):                                # each part is a subsequence of the plan (plan order kept)
        it = iter(key(o) for o in b.ops)
        assert all(any(k == kk for kk in it) for k in (key(o) for o in part))
    tail_kinds = {int(o["kind"]) for o in b.tail_ops}
    assert tail_kinds <= {_ffi.OP_WGRAD, _ffi.OP_WGRAD_GROUP, _ffi.OP_WGRAD_REDUCE_BATCH, _ffi.OP_CHAN_SUM_FINALIZE, _ffi.OP_BWD_REDUCE}
    assert _ffi.OP_WGRAD in tail_kinds and int(b.tail_ops[-1]["kind"]) == _ffi.OP_WGRAD_REDUCE_BATCH
    main_kinds = {int(o["kind"]) for o in b.main_ops}
    assert not (main_kinds & {_ffi.OP_WGRAD, _ffi.OP_WGRAD_GROUP, _ffi.OP_WGRAD_REDUCE_BATCH, _ffi.OP_CHAN_SUM_FINALIZE})
    assert int(b.main_ops[0]["kind"]) == _ffi.OP_ZERO                    # the gradient buffer is cleared before either part writes into it
    for k, o in enumerate(b.tail_ops):
        if int(o["kind"]) == _ffi.OP_BWD_REDUCE:                         # only as the reduction of a channel sum, directly in front of its finalize
            assert int(o["i"][0]) == 2 and int(b.tail_ops[k + 1]["kind"]) == _ffi.OP_CHAN_SUM_FINALIZE
    # an unsplit compile of the same view carries no parts
    assert compile_backward(net, view, 2, mask, name != "image_encoder").main_ops is None
