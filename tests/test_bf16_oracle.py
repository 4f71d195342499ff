"""Pins the bf16 plan emulation (oracle/bf16_plan.py: the checker of BASELINE config 3's forward AND backward) on the CPU:
  1. with every rounding switched off, its explicit backward (BatchNorm-backward in A*g + B*x + C form, pooled 4x4 data gradient of the
     nearest-upsample blocks, 2x2 phase forward, role-swapped ConvTranspose2d weight gradient, accumulated block-input gradients ...)
     must equal torch autograd of the plain fp64 network: the formulas are right independently of any GPU;
  2. with the roundings on, its forward must be the forward-only emulation that round 2 pinned (`bf16_rounding_points()`), and its
     gradients must sit next to the fp32 gradients at bf16 distance."""
import pytest
import torch

from oracle import bf16_plan as P
from oracle import ref_cpu as O

torch.set_num_threads(8)
NET_INPUT = {"image_encoder": (1, 32, 32), "shape_encoder": (4, 32, 48), "segmentation_decoder": (128, 2, 2),
             "shape_decoder": (128, 2, 3), "image_decoder": (128, 2, 2)}


def _net(name, golden_sd, dtype):
    net = O.build_networks(init=False)[name]
    net.load_state_dict(golden_sd[name])
    return net.to(dtype)


def _inputs(name, dtype, n=3):
    c, h, w = NET_INPUT[name]
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(n, c, h, w, generator=g)) if "decoder" in name else torch.rand(n, c, h, w, generator=g)
    return x.to(dtype)


@pytest.mark.parametrize("mode", ["A", "B"])
@pytest.mark.parametrize("name", list(NET_INPUT))
def test_unrounded_plan_backward_equals_autograd(name, mode, golden_sd):
    net = _net(name, golden_sd, torch.float64)
    x = _inputs(name, torch.float64).requires_grad_(True)
    ref_net = _net(name, golden_sd, torch.float64)
    P.ROUND = False
    try:
        if mode == "B":
            with O.bn_no_track(net), O.bn_no_track(ref_net):
                outs, rec = P.net_forward(net, x.detach())
                yo = ref_net(x)
                yo = yo if isinstance(yo, tuple) else (yo,)
                douts = [torch.randn(o.shape, generator=torch.Generator().manual_seed(3 + i), dtype=torch.float64) for i, o in enumerate(outs)]
                gin = torch.autograd.grad(sum((a * b).sum() for a, b in zip(yo, douts)), [x] + [p for p in ref_net.parameters() if p.requires_grad])
                names = [n for n, p in ref_net.named_parameters() if p.requires_grad]
                dx, grads = P.net_backward(net, rec, douts)
        else:
            outs, rec = P.net_forward(net, x.detach())
            yo = ref_net(x)
            yo = yo if isinstance(yo, tuple) else (yo,)
            douts = [torch.randn(o.shape, generator=torch.Generator().manual_seed(3 + i), dtype=torch.float64) for i, o in enumerate(outs)]
            gin = torch.autograd.grad(sum((a * b).sum() for a, b in zip(yo, douts)), [x] + list(ref_net.parameters()))
            names = [n for n, _ in ref_net.named_parameters()]
            dx, grads = P.net_backward(net, rec, douts)
    finally:
        P.ROUND = True
    for a, b in zip(outs, yo):
        assert float((a - b).abs().max()) <= 1e-10 * max(1.0, float(b.abs().max()))
    assert float((dx - gin[0]).abs().max()) <= 1e-9 * max(1.0, float(gin[0].abs().max())), "input gradient"
    for n, g in zip(names, gin[1:]):
        assert n in grads, n
        assert float((grads[n] - g).abs().max()) <= 1e-9 * max(1.0, float(g.abs().max())), n
    if mode == "A":                                            # running statistics moved exactly like torch's
        for (n, b), (_, br) in zip(net.named_buffers(), ref_net.named_buffers()):
            assert torch.allclose(b.double(), br.double(), rtol=1e-12, atol=1e-12), n


@pytest.mark.parametrize("name", list(NET_INPUT))
def test_rounded_plan_forward_is_the_round2_emulation_and_gradients_stay_close(name, golden_sd):
    net, net_old, net32 = (_net(name, golden_sd, torch.float32) for _ in range(3))
    x = _inputs(name, torch.float32, n=4)
    outs, rec = P.net_forward(net, x)
    with torch.no_grad(), O.bf16_rounding_points():
        yo = net_old(x)
    yo = yo if isinstance(yo, tuple) else (yo,)
    # Same rounding points, different fp32 summation details (statistics in fp64 here, torch's fp32 mean / var there): an operand that
    # lands on the other side of a bf16 rounding boundary moves by 2^-8 relative, and the training-mode BatchNorms of these randomly
    # initialised networks amplify it.  Measured between the two CPU emulations: up to 2e-2 of max|.| (max), 1.6e-3 (mean) -- this is
    # the noise floor ANY two implementations of the same bf16 computation see, and the yardstick for the GPU comparison.
    for a, b in zip(outs, yo):
        scale = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 4e-2 * scale and float((a - b).abs().mean()) <= 4e-3 * scale, "forward emulations disagree"
    for (n, b), (_, bo) in zip(net.named_buffers(), net_old.named_buffers()):
        assert torch.allclose(b.float(), bo.float(), rtol=2e-2, atol=2e-3), n
    # gradients: bf16 distance from fp32 autograd (loose: the point is that nothing is O(1) wrong), through the autograd wrapper
    xg = x.clone().requires_grad_(True)
    net2 = _net(name, golden_sd, torch.float32)
    with O.bf16_rounding_points(backward=True):
        y = net2(xg)
    y = y if isinstance(y, tuple) else (y,)
    douts = [torch.randn(o.shape, generator=torch.Generator().manual_seed(5 + i)) for i, o in enumerate(y)]
    sum((a * b).sum() for a, b in zip(y, douts)).backward()
    x32 = x.clone().requires_grad_(True)
    y32 = net32(x32)
    y32 = y32 if isinstance(y32, tuple) else (y32,)
    sum((a * b).sum() for a, b in zip(y32, douts)).backward()
    cos = lambda a, b: float((a.flatten().double() @ b.flatten().double()) / (a.double().norm() * b.double().norm()).clamp_min(1e-30))
    assert cos(xg.grad, x32.grad) > 0.95
    for (n, p), (_, q) in zip(net2.named_parameters(), net32.named_parameters()):
        if n.endswith(("conv.0.bias", "conv.3.bias", "inc.0.bias", "inc.3.bias", "final_conv.0.bias", "code_decoupler.0.bias", "code_decoupler.3.bias")):
            continue                                           # bias in front of a training-mode BatchNorm: true gradient 0
        assert p.grad is not None and cos(p.grad, q.grad) > 0.9, (n, cos(p.grad, q.grad))
