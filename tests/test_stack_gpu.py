"""Round 6: the standard and the hard-example pass of a network run their backward as ONE launch chain (nets.PassStack, solver.stack_passes).

The forward is untouched, so losses, masks and BatchNorm buffers must be BIT FOR BIT those of the per-pass step; the gradients differ by
the summation order only (one weight-gradient contraction over both passes instead of two partial results added; BatchNorm-backward sums
per group over another launch grid): checked per network against the per-pass backward at 2e-5 of the tensor's largest element, and the
per-pass BatchNorm mode (standard = A, hard = B: gamma / beta frozen) against a hand-made expectation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cooperative_training_and_latent_space_data_augmentation_amd import nets  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.autograd import net_apply  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.model_util import _disable_tracking_bn_stats  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.nets import PassStack  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.solver import STACK_ALL_FTN, AdvancedTripletReconSegmentationModel  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
from test_engine_gpu import NET_INPUT, _overrides, _solver, close, dev  # noqa: E402

DEV = "cuda"


def rel_close(a, b, tol, what):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: {err:.3e} of max|ref| {scale:.3e} (tolerance {tol:.1e})"


def _two_passes(net, xs, modes, douts_of, stacked, groups=1):
    """Forward of two passes (BatchNorm modes `modes`), then backward with the output gradients douts_of(outs); stacked or pass by pass.
    Returns (outputs, input gradients, flat parameter gradient)."""
    net.train()
    net.zero_grad()
    net._defer_grads, net._deferred, net._pending_bwd = True, [], 0
    net._stack = PassStack(net, 2) if stacked else None
    outs, leaves = [], []
    try:
        for x, mode in zip(xs, modes):
            x = x.clone().requires_grad_(True)
            leaves.append(x)
            if mode == "B":
                with _disable_tracking_bn_stats(net):
                    outs.append(net_apply(net, x, groups))
            else:
                outs.append(net_apply(net, x, groups))
        loss = sum((o * d).sum() for po, pd in zip(outs, [douts_of(o) for o in outs]) for o, d in zip(po, pd))
        loss.backward()
        if stacked:
            st = net._stack
            assert st.filled == 2
            cont = net.backward_stack(st)
            torch.autograd.backward([r for r, _ in cont], [g for _, g in cont])
        net.collect_deferred_grads()
        torch.cuda.synchronize()
        return ([tuple(o.detach().clone() for o in po) for po in outs], [x.grad.detach().clone() for x in leaves], net._flat.grad.detach().clone())
    finally:
        net._stack, net._defer_grads, net._deferred = None, False, []


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("modes", [("A", "B"), ("A", "A"), ("B", "B")])
@pytest.mark.parametrize("name", ["image_encoder", "segmentation_decoder", "image_decoder", "shape_encoder"])
def test_stacked_backward_of_two_passes_equals_the_two_per_pass_backwards(golden_sd, name, modes, dtype):
    c, h, w = NET_INPUT[name]
    n = 4
    g = torch.Generator().manual_seed(sum(map(ord, name + modes[0] + modes[1])))
    xs = [dev(torch.rand(n, c, h, w, generator=g) if name.endswith("encoder") else torch.randn(n, c, h, w, generator=g).relu()) for _ in range(2)]
    results = []
    for stacked in (False, True):
        net = nets.build_networks(device=DEV, state_dicts={name: golden_sd[name]}, dtype=dtype)[name]
        buf0 = net._bflat.clone()
        cnt = [0]

        def d_of(outs, cnt=cnt):
            res = [dev(torch.randn(o.shape, generator=torch.Generator().manual_seed(100 + 10 * cnt[0] + k))) for k, o in enumerate(outs)]
            cnt[0] += 1
            return res
        results.append(_two_passes(net, xs, modes, d_of, stacked) + (net._bflat.clone(), net._nbt.clone()))
        assert ("A" in modes) == (not torch.equal(buf0, net._bflat))
    (o0, dx0, g0, b0, t0), (o1, dx1, g1, b1, t1) = results
    for pa, pb in zip(o0, o1):
        for a, b in zip(pa, pb):
            assert torch.equal(a, b), "the slot forward must be the per-pass forward bit for bit"
    assert torch.equal(b0, b1) and torch.equal(t0, t1), "BatchNorm running statistics"
    tol = 2e-5 if dtype == "fp32" else 2e-2         # (bf16: intermediate gradients are bf16-rounded; another statistics order moves roundings)
    for p, (a, b) in enumerate(zip(dx0, dx1)):
        rel_close(b, a, tol, f"input gradient of pass {p}")
    rel_close(g1, g0, tol if dtype == "fp32" else 3e-2, "flat parameter gradient")
    # mode B passes add nothing to gamma / beta: with ("B", "B") their gradient stays exactly zero
    if modes == ("B", "B"):
        for key, bn in net._bns.items():
            assert float(g1[bn.g_off:bn.g_off + bn.c].abs().max()) == 0.0 and float(g1[bn.b_off:bn.b_off + bn.c].abs().max()) == 0.0, key


@pytest.mark.parametrize("case", ["C_step_channel_spatial", "D_step_dropout"])
def test_cooperative_step_with_stacked_backward_equals_per_pass_backward(golden_cases, golden_sd, case):
    C = golden_cases[case]
    ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    clean, label, noisy = dev(C["clean"]), dev(C["label"]), dev(C["noisy"])
    got = {}
    for stacked in (False, True):
        s = _solver(golden_sd)
        s.stack_passes = STACK_ALL_FTN if stacked else ()
        losses = s.cooperative_step(clean, label, noisy, C["img_cfg"], C["seg_cfg"], image_override=ov_img, seg_override=ov_seg, do_optim=False)
        torch.cuda.synchronize()
        got[stacked] = (torch.stack([v.float() for v in losses]).cpu(), {k: m._flat.grad.detach().clone() for k, m in s.model.items()},
                        {k: (m._bflat.clone(), m._nbt.clone()) for k, m in s.model.items()}, dict(s.last_masks))
    assert torch.equal(got[True][0], got[False][0]), "the 8 losses (forward untouched)"
    assert torch.allclose(got[True][0].double(), C["losses"], atol=1e-4, rtol=0)
    for k in got[True][2]:
        assert torch.equal(got[True][2][k][0], got[False][2][k][0]) and torch.equal(got[True][2][k][1], got[False][2][k][1]), f"BatchNorm buffers of {k}"
    for k in got[True][3]:
        assert torch.equal(got[True][3][k], got[False][3][k]), f"mask {k}"
    for k in got[True][1]:
        rel_close(got[True][1][k], got[False][1][k], 5e-5, f"gradient of {k}")


def test_stacked_step_takes_fewer_launches_and_keeps_training_state_deterministic(golden_cases, golden_sd):
    from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib
    C = golden_cases["D_step_dropout"]
    ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    clean, label, noisy = dev(C["clean"]), dev(C["label"]), dev(C["noisy"])
    counts, states = {}, []
    for stacked in (False, True, True):
        s = _solver(golden_sd)
        s.stack_passes = STACK_ALL_FTN if stacked else ()
        s.cooperative_step(clean, label, noisy, C["img_cfg"], C["seg_cfg"], image_override=ov_img, seg_override=ov_seg)      # (plans compiled, arenas leased)
        torch.cuda.synchronize()
        c0 = lib.ctl_launch_count()
        s.cooperative_step(clean, label, noisy, C["img_cfg"], C["seg_cfg"], image_override=ov_img, seg_override=ov_seg)
        torch.cuda.synchronize()
        counts.setdefault(stacked, lib.ctl_launch_count() - c0)
        if stacked:
            states.append({k: m._flat_data.detach().clone() for k, m in s.model.items()})
    assert counts[True] < counts[False] - 100, counts
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), f"two runs of the stacked step must agree bit for bit ({k})"


def test_stacked_step_under_graph_capture_replays_the_eager_stacked_step(golden_sd):
    """hipGraph capture of the stacked step (every stacked backward on the capture's origin stream, no tail split: see solver._backward_stacks):
    the same launches as the eager stacked step, so weights and losses agree bit for bit after two steps."""
    from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
    drop_i = {"loss_name": "mse", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    drop_s = {"loss_name": "ce", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(4, 64, 64, seed=7))
    keeps = [dev((torch.rand(4, 128, generator=torch.Generator().manual_seed(k)) > 0.5).float()) for k in (1, 2)]
    ovi, ovs = {"keep": keeps[0]}, {"keep": keeps[1]}
    a, b = _solver(golden_sd), _solver(golden_sd)
    a.stack_passes = b.stack_passes = STACK_ALL_FTN
    for _ in range(2):
        la = a.cooperative_step(clean, label, noisy, drop_i, drop_s, image_override=ovi, seg_override=ovs)
    g = CooperativeStepGraph(b, drop_i, drop_s)
    orig = g._run_step

    def run(schemes, do_optim, hook):          # (the injected keep patterns: the graph's own draws come from its device RNG state)
        c, l, n = g.static_in
        return b.cooperative_step(c, l, n, drop_i, drop_s, image_override=ovi, seg_override=ovs, do_optim=do_optim, grad_hook=hook)
    g._run_step = run
    for _ in range(2):
        lb = g(clean, label, noisy)
    torch.cuda.synchronize()
    assert all(torch.equal(u.detach(), v.detach()) for u, v in zip(la, lb))
    for k in a.model:
        assert torch.equal(a.model[k]._flat_data, b.model[k]._flat_data), k
