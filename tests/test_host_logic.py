"""CPU tests of the host side: reference-compatible state dicts / init, flat parameter storage, plan compilation,
optimizer-state format.  No kernel is launched."""
import numpy as np
import pytest
import torch

from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, init, nets
from cooperative_training_and_latent_space_data_augmentation_amd.optim import FlatAdam


def test_seed0_init_matches_reference(golden_sd):
    torch.manual_seed(0)
    sd = init.reference_init_state_dicts()
    for k in golden_sd:
        assert list(sd[k].keys()) == list(golden_sd[k].keys())
        assert all(torch.equal(sd[k][n], golden_sd[k][n]) for n in sd[k]), k


def test_state_dict_round_trip_and_flat_views(golden_sd):
    model = nets.build_networks(device="cpu", state_dicts=golden_sd)
    for k, net in model.items():
        sd = net.state_dict()
        assert list(sd.keys()) == list(golden_sd[k].keys())
        assert all(torch.equal(sd[n], golden_sd[k][n]) for n in sd)
        # every named parameter / gradient is a view of the flat buffers, 256-byte aligned
        for n, p in net.named_parameters():
            off = net._poff[n]
            assert off % 64 == 0 and p.data_ptr() == net._flat_data.data_ptr() + 4 * off
            assert p.grad.data_ptr() == net._flat.grad.data_ptr() + 4 * off
        net.zero_grad()
        assert all(p.grad is not None for p in net.parameters())      # zero_grad keeps the views (upstream calls it)
    n_params = {k: sum(p.numel() for p in m.parameters()) for k, m in model.items()}
    assert n_params == {"image_encoder": 1125888, "segmentation_decoder": 161732, "shape_encoder": 830640,
                        "shape_decoder": 161732, "image_decoder": 248961}


def test_plans_compile_for_all_modes():
    model = nets.build_networks(device="cpu")
    enc, dec, dimg = model["image_encoder"], model["segmentation_decoder"], model["image_decoder"]
    for mode in "ABC":
        p = enc._compile_forward(2, 48, 64, mode)
        assert p.out_shapes == [(2, 3, 4, 128)] * 2
        kinds = [int(o["kind"]) for o in p.ops]
        assert (_ffi.OP_BN_EVAL in kinds) == (mode == "C") and (_ffi.OP_BN_FINALIZE in kinds) == (mode != "C")
        if mode != "C":
            upd = [int(o["i"][2]) for o in p.ops if int(o["kind"]) == _ffi.OP_BN_FINALIZE]
            assert all(u == (1 if mode == "A" else 0) for u in upd) and len(upd) == 13       # 11 + 2 BatchNorm layers
    f = dec._compile_forward(2, 3, 4, "A")
    assert f.out_shapes == [(2, 48, 64, 4)]
    full = dec._compile_backward(f, "A", (True,), True, True, True)
    dgrad_only = dec._compile_backward(f, "A", (True,), True, False, False)          # the saliency pass: no wgrad at all
    kinds = [int(o["kind"]) for o in dgrad_only.ops]
    assert _ffi.OP_WGRAD not in kinds and _ffi.OP_WGRAD_REDUCE not in kinds and _ffi.OP_ZERO not in kinds
    assert sum(int(o["kind"]) == _ffi.OP_WGRAD for o in full.ops) == 13               # 4 blocks x 3 convs + final conv
    fi = dimg._compile_forward(2, 4, 4, "A")
    bi = dimg._compile_backward(fi, "A", (True,), True, True, True)
    assert sum(int(o["kind"]) == _ffi.OP_WGRAD for o in bi.ops) == 17                 # + 4 ConvTranspose2d
    assert any(int(o["kind"]) == _ffi.OP_SIGMOID_BWD for o in bi.ops)


def test_deferred_weight_gradients_are_grouped_at_the_end_of_a_backward_plan():
    """nets.GROUP_WGRAD (round 5): the X3 weight gradients that can share a launch are deferred to the end of the backward plan and emitted as
    CTL_OP_WGRAD_GROUP records, each followed by its member WGRAD records; every member carries its pixel-split count (i[24]), its partial
    buffer is sized from that count, the reduction table holds the same count, and nothing but the reduction follows the groups."""
    model = nets.build_networks(device="cpu")
    enc = model["image_encoder"]
    f = enc._compile_forward(16, 256, 256, "A")
    b = enc._compile_backward(f, "A", (True, True), False, True, True)
    kinds = [int(o["kind"]) for o in b.ops]
    groups = [k for k, v in enumerate(kinds) if v == _ffi.OP_WGRAD_GROUP]
    assert groups, "a bs16 256^2 encoder backward has groupable weight gradients (32 -> 32 ... 128 -> 128 3x3 layers)"
    first = groups[0]
    assert _ffi.OP_CONV not in kinds[first:] and kinds[-1] == _ffi.OP_WGRAD_REDUCE_BATCH      # the groups are the tail of the plan
    table = b.table_np
    n_members = 0
    for k in groups:
        n = int(b.ops[k]["i"][0])
        assert 2 <= n <= _ffi.WGRAD_GROUP_MAX
        cls = set()
        for m in b.ops[k + 1:k + 1 + n]:
            assert int(m["kind"]) == _ffi.OP_WGRAD
            d = np.frombuffer(np.ascontiguousarray(m["i"][:nets.CONV_WORDS]).tobytes(), dtype=_ffi.CONV_DTYPE)[0]
            sp = int(m["i"][nets.CONV_WORDS])
            assert sp >= 1 and d["cin"] % 32 == 0 and d["cout"] % 32 == 0 and d["ks"] == 3 and d["stride"] == 1 and d["dt"] == _ffi.DT_X3
            cls.add((int(d["in_mode"]), int(m["slot"][6]) >= 0))
            rec = [r for r in table if int(r[0]) * 4 == int(m["off"][4])]          # the reduction record of this member's partial buffer
            assert len(rec) == 1 and int(rec[0][4]) == sp
        assert len(cls) == 1                                                       # one kernel instantiation per launch
        n_members += n
    assert n_members >= 8
    # the switch: without it every weight gradient is a launch of its own, in place
    nets.GROUP_WGRAD = False
    try:
        b0 = enc._compile_backward(f, "A", (True, True), False, True, True)
    finally:
        nets.GROUP_WGRAD = True
    k0 = [int(o["kind"]) for o in b0.ops]
    assert _ffi.OP_WGRAD_GROUP not in k0 and sum(v == _ffi.OP_WGRAD for v in k0) == sum(v == _ffi.OP_WGRAD for v in kinds)


def test_bf16_weight_gradients_are_stacked_by_kernel_instantiation():
    """nets.GROUP_WGRAD_BF16 (round 5): the bf16 family's weight gradients ride in stacked launches too -- class = 0x100 | the kernel instantiation,
    every member's split count comes from ctl_wgrad_group_plan (at most the count of a launch of its own) and sizes its partial buffer."""
    model = nets.build_networks(device="cpu", dtype="bf16")
    enc = model["image_encoder"]
    f = enc._compile_forward(16, 256, 256, "A")
    b = enc._compile_backward(f, "A", (True, True), False, True, True)
    kinds = [int(o["kind"]) for o in b.ops]
    groups = [k for k, v in enumerate(kinds) if v == _ffi.OP_WGRAD_GROUP]
    assert len(groups) >= 3 and kinds[-1] == _ffi.OP_WGRAD_REDUCE_BATCH
    for k in groups:
        n = int(b.ops[k]["i"][0])
        cls = set()
        for m in b.ops[k + 1:k + 1 + n]:
            assert int(m["kind"]) == _ffi.OP_WGRAD
            d = np.frombuffer(np.ascontiguousarray(m["i"][:nets.CONV_WORDS]).tobytes(), dtype=_ffi.CONV_DTYPE)[0]
            assert d["dt"] & _ffi.DT_BF16
            c = int(_ffi.lib.ctl_wgrad_group_class(_ffi.desc_ptr(np.atleast_1d(d)), 1 if int(m["slot"][6]) >= 0 else 0))
            assert c >= 0x100
            cls.add(c)
            sp = int(m["i"][nets.CONV_WORDS])
            assert 1 <= sp <= int(_ffi.lib.ctl_wgrad_splits(_ffi.desc_ptr(np.atleast_1d(d))))
            rec = [r for r in b.table_np if int(r[0]) * 4 == int(m["off"][4])]
            assert len(rec) == 1 and int(rec[0][4]) == sp
        assert len(cls) == 1
    nets.GROUP_WGRAD_BF16 = False
    try:
        b0 = enc._compile_backward(f, "A", (True, True), False, True, True)
    finally:
        nets.GROUP_WGRAD_BF16 = True
    assert _ffi.OP_WGRAD_GROUP not in [int(o["kind"]) for o in b0.ops]


def test_without_gpu_the_product_path_fails_loudly():
    model = nets.build_networks(device="cpu")
    x = torch.rand(1, 1, 32, 32)
    with pytest.raises(_ffi.CtlError):
        model["image_encoder"](x)


def test_flat_adam_state_dict_is_torch_compatible():
    net = nets.build_networks(device="cpu")["segmentation_decoder"]
    opt = FlatAdam(net, lr=1e-4)
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in net.parameters()], lr=1e-4)
    for p in ref.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    ref.step()
    opt.load_state_dict(ref.state_dict())                 # upstream `<name>_optim.pth` / snapshot format
    assert opt.step_count == 1
    sd = opt.state_dict()
    assert set(sd["state"].keys()) == set(ref.state_dict()["state"].keys())
    for i, st in ref.state_dict()["state"].items():
        assert torch.allclose(sd["state"][i]["exp_avg"], st["exp_avg"]) and torch.allclose(sd["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
    assert sd["param_groups"][0]["lr"] == 1e-4 and sd["param_groups"][0]["betas"] == (0.9, 0.999)
