"""Engine-level parity on MI355X: whole networks and whole cooperative steps (HIP plans through the C-ABI) against the
CPU oracle on the same seeded inputs, and against the golden vectors recorded from the real reference.
Tolerances: forward tensors / losses 1e-4 abs (north_star), gradients 5e-4 of the tensor's max."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, nets, ops  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.model_util import _disable_tracking_bn_stats  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel  # noqa: E402

DEV = "cuda"
torch.set_num_threads(8)


def dev(x):
    x = x.to(DEV)
    return x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x.contiguous()


def close(a, b, atol=1e-4, rel=0.0, what=""):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = atol + rel * float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= tol, f"{what}: max err {err:.3e} > tol {tol:.3e} (max|ref| {float(b.abs().max()):.3e})"


def is_dead_bias(name):
    """Bias of a conv that feeds a training-mode BatchNorm: its true gradient is exactly 0, every implementation
    (the reference included) returns rounding noise there."""
    return name.endswith(("conv.0.bias", "conv.3.bias", "inc.0.bias", "inc.3.bias", "final_conv.0.bias",
                          "code_decoupler.0.bias", "code_decoupler.3.bias"))


def grads_close_robust(a, b, what):
    """Gradient parity in the presence of LeakyReLU-derivative ties: a pre-activation within fp32 rounding of 0 takes
    slope 1 in one implementation and 0.2 in the other (verified with tools/debug_sign_ties.py), which perturbs the
    gradient downstream of that single element.  So: small relative L2 error overall and tight agreement in the bulk."""
    a, b = a.detach().cpu().float().flatten(), b.detach().cpu().float().flatten()
    scale = max(float(b.abs().max()), 1e-12)
    l2 = float((a - b).norm()) / max(float(b.norm()), 1e-12)
    q90 = float(torch.quantile((a - b).abs()[:1_000_000], 0.9))
    assert l2 <= 3e-2, f"{what}: relative L2 error {l2:.3e}"
    # 4e-3: the golden gradients are the reference's own fp32 run, itself 0.3-0.9 % away from fp64 on this network; a change of
    # the fp32 summation order in the kernels moves this figure by ~1e-3 (2.0e-3 and 2.05e-3 were both observed)
    assert q90 <= 4e-3 * scale + 2e-6, f"{what}: 90th-percentile abs error {q90:.3e} (max|ref| {scale:.3e})"


def min_preactivation(onet, x):
    """Smallest |input| of any (Leaky)ReLU in the oracle network for input x (tie detector)."""
    vals, hooks = [], []
    for m in onet.modules():
        if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU)):
            hooks.append(m.register_forward_hook(lambda mod, a, o: vals.append(float(a[0].detach().abs().min()))))
    if hasattr(onet, "inc") or hasattr(onet, "general_encoder"):
        enc = onet.general_encoder if hasattr(onet, "general_encoder") else onet
        hooks.append(enc.inc.register_forward_hook(lambda mod, a, o: vals.append(float(o.detach().abs().min()))))
    with torch.no_grad():
        onet(x)
    for h in hooks:
        h.remove()
    return min(vals)


def make_pair(name, golden_sd):
    onet = O.build_networks(init=False)[name]
    onet.load_state_dict(golden_sd[name])
    hnet = nets.build_networks(device=DEV, state_dicts={name: golden_sd[name]})[name]
    return onet, hnet


NET_INPUT = {"image_encoder": (1, 64, 48), "shape_encoder": (4, 48, 64), "segmentation_decoder": (128, 4, 3),
             "shape_decoder": (128, 3, 4), "image_decoder": (128, 4, 4)}


# plan-compiler switches of the fp32 backward (nets.py): the default (tail reduction in the launch that writes dOut, tail apply staged in
# its consumers), everything folded into the convs (in-block BatchNorm too: reduction in the data-gradient epilogue, apply staged), and
# every pass stand-alone -- each against the oracle
PLAN_SWITCHES = {"default": {}, "all_folded": {"FUSE_BNBWD": True, "FUSE_BNAPPLY": True, "FUSE_TAIL": True, "FUSE_PAIR": True},
                 "stand_alone_passes": {"FUSE_BNBWD": False, "FUSE_BNAPPLY": False, "FUSE_TAIL": False, "FUSE_PAIR": False},
                 "staged_apply_without_tail_epilogue": {"FUSE_BNAPPLY": True, "FUSE_TAIL": False},
                 "blocks_folded_head_pairs_not": {"FUSE_PAIR": False},
                 "virtual_tensors_written_by_the_data_gradients": {"FUSE_XOUT": True}}


@pytest.mark.parametrize("switches", list(PLAN_SWITCHES))
@pytest.mark.parametrize("mode", ["A", "B"])
@pytest.mark.parametrize("name", list(NET_INPUT))
def test_network_forward_backward_vs_oracle(name, mode, switches, golden_sd):
    old = {k: getattr(nets, k) for k in PLAN_SWITCHES[switches]}
    for k, v in PLAN_SWITCHES[switches].items():
        setattr(nets, k, v)
    try:
        _network_forward_backward_vs_oracle(name, mode, golden_sd)
    finally:
        for k, v in old.items():
            setattr(nets, k, v)


def _network_forward_backward_vs_oracle(name, mode, golden_sd):
    onet, hnet = make_pair(name, golden_sd)
    onet.train()
    hnet.train()
    c, h, w = NET_INPUT[name]
    for seed in range(64):                 # first input without an activation tie (see grads_close_robust)
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(3, c, h, w, generator=g)
        if "decoder" in name:
            x = torch.relu(torch.randn(3, c, h, w, generator=g))
        buf = {k: v.clone() for k, v in onet.state_dict().items()}
        ok = min_preactivation(onet, x) > 4e-6
        onet.load_state_dict(buf)          # the probe moved the running statistics
        if ok:
            break
    else:
        pytest.skip("no tie-free input found")
    xo = x.clone().requires_grad_(True)
    xh = dev(x).requires_grad_(True)
    if mode == "B":
        with O.bn_no_track(onet):
            yo = onet(xo)
        with _disable_tracking_bn_stats(hnet):
            yh = hnet(xh)
    else:
        yo, yh = onet(xo), hnet(xh)
    yo = yo if isinstance(yo, tuple) else (yo,)
    yh = yh if isinstance(yh, tuple) else (yh,)
    douts = []
    for a, b in zip(yh, yo):
        close(a, b, what=f"{name} output")
        douts.append(torch.randn(b.shape, generator=g))
    torch.autograd.backward(yo, douts)
    torch.autograd.backward(yh, [dev(d) for d in douts])
    close(xh.grad, xo.grad, atol=1e-6, rel=5e-4, what=f"{name} dx")
    hp = dict(hnet.named_parameters())
    for n, p in onet.named_parameters():
        if mode == "B" and p.grad is None:             # gamma/beta are frozen in mode B
            assert float(hp[n].grad.abs().max()) == 0.0, n
            continue
        if is_dead_bias(n):
            wn = float(dict(onet.named_parameters())[n[:-4] + "weight"].grad.norm())
            assert float(hp[n].grad.abs().max()) <= 1e-3 * wn + 1e-4, n
            continue
        close(hp[n].grad, p.grad, atol=2e-6, rel=5e-4, what=f"{name} grad {n}")
    hb = dict(hnet.named_buffers())
    for n, b in onet.named_buffers():
        close(hb[n].double(), b.double(), atol=1e-5, rel=1e-5, what=f"{name} buffer {n}")
    if mode == "B":
        assert int(hb[[k for k in hb if k.endswith("num_batches_tracked")][0]]) == 0


@pytest.mark.parametrize("mode", ["A", "B"])
@pytest.mark.parametrize("name,n", [("shape_encoder", 3), ("shape_decoder", 2), ("image_encoder", 16), ("segmentation_decoder", 5)])
def test_grouped_pass_equals_consecutive_passes(name, n, mode, golden_sd):
    """One pass over two stacked batches with BatchNorm groups (ctl_conv.groups = 2) == the two passes one after the other:
    outputs, input gradients, accumulated parameter gradients, running statistics (updated in call order)."""
    from cooperative_training_and_latent_space_data_augmentation_amd.autograd import net_apply
    c, h, w = NET_INPUT[name]
    if name == "image_encoder":
        h, w = 96, 80                       # several tiles per image and 16 images: blocks walk across the group boundary
    if "decoder" in name:
        h, w = 4 * h, 4 * w                 # enough pixels per BatchNorm group for a well-conditioned backward
    g = torch.Generator().manual_seed(11)
    xa, xb = torch.rand(n, c, h, w, generator=g), torch.rand(n, c, h, w, generator=g) * 1.7 - 0.2
    nets_ = [nets.build_networks(device=DEV, state_dicts={name: golden_sd[name]})[name] for _ in range(2)]
    res = []
    for net, grouped in zip(nets_, (False, True)):
        net.train()
        xs = [dev(xa).requires_grad_(True), dev(xb).requires_grad_(True)]
        ctx = _disable_tracking_bn_stats(net) if mode == "B" else None
        if ctx is not None:
            ctx.__enter__()
        if grouped:
            outs = net_apply(net, torch.cat(xs, 0), groups=2)
            outs = [tuple(o[:n] for o in outs), tuple(o[n:] for o in outs)]
        else:
            outs = [net_apply(net, xs[0]), net_apply(net, xs[1])]
        if ctx is not None:
            ctx.__exit__(None, None, None)
        gg = torch.Generator().manual_seed(5)
        douts = [[torch.randn(o.shape, generator=gg) for o in oo] for oo in outs]
        torch.autograd.backward([o for oo in outs for o in oo], [dev(d) for dd in douts for d in dd])
        res.append((outs, [x.grad for x in xs], {k: v.grad.clone() for k, v in net.named_parameters()},
                    {k: v.clone() for k, v in net.named_buffers()}))
    (o1, dx1, g1, b1), (o2, dx2, g2, b2) = res
    for oa, ob in zip(o1, o2):
        for a, b in zip(oa, ob):
            close(b, a, atol=2e-5, rel=1e-5, what=f"{name} grouped output")
    # gradients: robust comparison (a LeakyReLU input within rounding of 0 may take the other slope when the order of the
    # statistics partial sums changes -- see grads_close_robust)
    for a, b in zip(dx1, dx2):
        grads_close_robust(b, a, f"{name} grouped dx")
    for k in g1:
        if is_dead_bias(k):
            continue
        grads_close_robust(g2[k], g1[k], f"{name} grouped grad {k}")
    for k in b1:
        close(b2[k].double(), b1[k].double(), atol=1e-6, rel=1e-6, what=f"{name} grouped buffer {k}")
    nbt = [k for k in b2 if k.endswith("num_batches_tracked")][0]
    assert int(b2[nbt]) == (2 if mode == "A" else 0)


@pytest.mark.parametrize("name", list(NET_INPUT))
def test_network_eval_mode_vs_oracle(name, golden_sd):
    onet, hnet = make_pair(name, golden_sd)
    c, h, w = NET_INPUT[name]
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, c, h, w, generator=g)
    with torch.no_grad():
        for _ in range(2):                     # move the running statistics first
            onet(x * 1.5)
            hnet(dev(x * 1.5))
        onet.eval()
        hnet.eval()
        yo, yh = onet(x), hnet(dev(x))
    for a, b in zip(yh if isinstance(yh, tuple) else (yh,), yo if isinstance(yo, tuple) else (yo,)):
        close(a, b, what=f"{name} eval output")


def _solver(golden_sd):
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    for k, m in s.model.items():
        m.load_state_dict(golden_sd[k])
    return s


def _overrides(rec, cfgs):
    ovs, draws, noises, keeps = [], list(rec["rand_draws"]), list(rec["soft_noises"]), list(rec["dropout_keeps"])
    for cfg, shp in zip(cfgs, (rec["z_i"].shape, rec["z_s"].shape)):
        ov = {}
        if cfg["mask_type"] == "dropout":
            ov["keep"] = keeps.pop(0).to(DEV)
        else:
            L = shp[1] if cfg["mask_type"] == "channel" else shp[2] * shp[3]
            if cfg["random_threshold"]:
                ov["k"] = int(L * (draws.pop(0) * cfg["max_threshold"]))
            if cfg["if_soft"]:
                ov["soft_noise"] = noises.pop(0).to(DEV)
        ovs.append(ov)
    return ovs


def test_standard_training_vs_golden(golden_cases, golden_sd):
    A = golden_cases["A_standard"]
    s = _solver(golden_sd)
    s.train()
    s.reset_all_optimizers()
    std = s.standard_training(dev(A["clean"]), dev(A["label"]), dev(A["noisy"]))
    (std[0] + std[1] + std[2] + std[3]).backward()
    got = torch.stack([v.detach() for v in std]).cpu().double()
    assert torch.allclose(got, A["losses"], atol=1e-4, rtol=0), (got, A["losses"])
    close(s.z_i, A["z_i"], what="z_i")
    close(s.z_s, A["z_s"], what="z_s")
    for key, gref in A["grads"].items():
        k, n = key.split("/")
        if not is_dead_bias(n):
            grads_close_robust(dict(s.model[k].named_parameters())[n].grad, gref, key)
    for key, b in A["buffers_after"].items():
        k, n = key.split("/")
        close(dict(s.model[k].named_buffers())[n].double(), b.double(), atol=1e-5, rel=1e-5, what=key)


def _to64(ov):
    return {k: ((v.cpu().double() if v.is_floating_point() else v.cpu()) if torch.is_tensor(v) else v) for k, v in ov.items()}


@pytest.mark.parametrize("case", ["C_step_channel_spatial", "D_step_dropout", "E_step_soft_random"])
def test_full_cooperative_step_vs_golden(golden_cases, golden_sd, case):
    """One whole iteration (standard + generation + hard + backward + Adam) against the reference's recorded run.
    Forward quantities (8 losses, masks, BN buffers) are compared strictly.  Gradients of this network are
    ill-conditioned in fp32 (LeakyReLU-derivative ties: the REFERENCE's fp32 gradients are themselves 0.3-1 % away
    from an fp64 run of the same step), so the yardstick is the fp64 oracle: the HIP error must stay within a small
    factor of the reference's own fp32 error."""
    C = golden_cases[case]
    s = _solver(golden_sd)
    ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    losses = s.cooperative_step(dev(C["clean"]), dev(C["label"]), dev(C["noisy"]), C["img_cfg"], C["seg_cfg"],
                                image_override=ov_img, seg_override=ov_seg)
    got = torch.stack([v.detach().float() for v in losses]).cpu().double()
    assert torch.allclose(got, C["losses"], atol=1e-4, rtol=0), (got, C["losses"])
    if C["img_cfg"]["mask_type"] != "dropout":
        assert torch.equal(s.last_masks["image"].cpu(), C["masks"][0])       # integer-exact selection
        assert torch.equal(s.last_masks["seg"].cpu(), C["masks"][1])
    o64 = O.OracleSolver(state_dicts=golden_sd).double()
    l64 = o64.cooperative_step(C["clean"].double(), C["label"], C["noisy"].double(), C["img_cfg"], C["seg_cfg"],
                               image_override=_to64(ov_img), seg_override=_to64(ov_seg), do_optim=False)
    assert float((got - torch.tensor(l64, dtype=torch.float64)).abs().max()) < 1e-4
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    e_hip, e_ref = {}, {}
    for key, gref in C["grads"].items():
        k, n = key.split("/")
        if is_dead_bias(n):
            continue
        g64 = dict(o64.model[k].named_parameters())[n].grad
        e_hip[key] = rel(dict(s.model[k].named_parameters())[n].grad.detach().cpu().double(), g64)
        e_ref[key] = rel(gref.double(), g64)
    noise = max(e_ref.values())                        # fp32 noise level of this step, measured on the reference itself
    for key in e_hip:
        assert e_hip[key] <= max(5 * noise, 1e-2), f"{key}: HIP-vs-fp64 {e_hip[key]:.2e}, reference-vs-fp64 {e_ref[key]:.2e}"
    bad = []
    for k, m in s.model.items():                       # every parameter: gradient norm within 5 % of the fp64 norm
        for n, p in m.named_parameters():
            if is_dead_bias(n):
                continue
            n64 = float(dict(o64.model[k].named_parameters())[n].grad.norm())
            nh = float(p.grad.detach().double().norm())
            if abs(nh - n64) > 5e-2 * n64 + 1e-7:
                bad.append((k, n, nh, n64))
    assert not bad, bad[:4]
    for key, b in C["buffers_after"].items():
        k, n = key.split("/")
        close(dict(s.model[k].named_buffers())[n].double(), b.double(), atol=2e-5, rel=1e-5, what=key)
    for key, p in C["params_after"].items():          # Adam's first step is +-lr (see tests/test_oracle_golden.py)
        k, n = key.split("/")
        close(dict(s.model[k].named_parameters())[n], p, atol=2.1e-4, what=key)

@pytest.mark.parametrize("variant", ["both", "image_only", "seg_only", "no_cfgs", "no_latent_DA", "separate_training", "targeted_C", "targeted_E"])
def test_two_stream_step_is_bitwise_identical(golden_cases, golden_sd, variant):
    """The iteration is issued as two launch chains on two HIP streams (solver.two_streams).  Every kernel is deterministic, so
    two training steps must leave bit-identical weights, losses and BatchNorm buffers with and without it -- a race would show up
    here.  targeted_C / targeted_E: channel / spatial masks, whose saliency pass is a third tracking pass of the image decoder."""
    C = golden_cases[{"targeted_C": "C_step_channel_spatial", "targeted_E": "E_step_soft_random"}.get(variant, "D_step_dropout")]
    outs = []
    for two in (False, True, True):
        s = _solver(golden_sd)
        s.two_streams = two
        if two and s._side is None:
            s._side = torch.cuda.Stream(device=s.device)
        ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
        kw = dict(img_cfg=C["img_cfg"], seg_cfg=C["seg_cfg"], image_override=ov_img, seg_override=ov_seg)
        if variant == "image_only":
            kw.update(seg_cfg=None, seg_override=None)
        elif variant == "seg_only":
            kw.update(img_cfg=None, image_override=None)
        elif variant == "no_cfgs":        # cooperative_step's defaults: neither code perturbed, the hard branch is four constant zeros
            kw.update(img_cfg=None, seg_cfg=None, image_override=None, seg_override=None)      # (the side-chain sweep had nothing to differentiate)
        elif variant == "no_latent_DA":
            kw.update(latent_DA=False)
        elif variant == "separate_training":
            kw.update(separate_training=True)
        for _ in range(3 if variant.startswith("targeted") else 2):
            losses = s.cooperative_step(dev(C["clean"]), dev(C["label"]), dev(C["noisy"]), **kw)
        torch.cuda.synchronize()
        outs.append((torch.stack([v.detach().float() for v in losses]).cpu(),
                     {k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()},
                     {k: (m._bflat.detach().cpu().clone(), m._nbt.detach().cpu().clone()) for k, m in s.model.items()}))
    for l, w, b in outs[1:]:
        assert torch.equal(l, outs[0][0])
        for k in w:
            assert torch.equal(w[k], outs[0][1][k]), k
            # BatchNorm running statistics too: the image decoder's are written on both chains, in an event-enforced order
            assert torch.equal(b[k][0], outs[0][2][k][0]) and torch.equal(b[k][1], outs[0][2][k][1]), k



def test_predict_vs_golden(golden_cases, golden_sd):
    F_ = golden_cases["F_predict"]
    s = _solver(golden_sd)
    s.train()
    with torch.no_grad():
        for i in range(3):
            c_, l_, n_ = O.synthetic_batch(2, 64, 64, seed=10 + i, structured=True)
            s.standard_training(dev(c_), dev(l_), dev(n_))
    for key, b in F_["buffers_after"].items():
        k, n = key.split("/")
        close(dict(s.model[k].named_buffers())[n].double(), b.double(), atol=2e-5, rel=1e-5, what=key)
    p1 = s.predict(dev(F_["vol"]), n_iter=1)
    p2 = s.predict(dev(F_["vol"]), n_iter=2)
    close(p1, F_["logits_n1"], atol=1e-4, rel=2e-5, what="logits n_iter=1")      # |logit| up to ~7: 1e-4 abs + 2e-5 rel
    close(p2, F_["logits_n2"], atol=1e-4, rel=2e-5, what="logits n_iter=2")
    for p, key in ((p1, "argmax_n1"), (p2, "argmax_n2")):
        ref_logits = F_["logits_n1" if key.endswith("1") else "logits_n2"]
        top2 = ref_logits.topk(2, dim=1)[0]
        safe = (top2[:, 0] - top2[:, 1]) > 1e-3
        lab = ops.argmax_c(p).cpu()
        assert torch.equal(lab[safe], F_[key][safe])            # integer label maps bit-exact away from near-ties
        for cls in range(1, 4):
            d_h, d_r = O.dice(lab.numpy() == cls, F_["vlab"].numpy() == cls), O.dice(F_[key].numpy() == cls, F_["vlab"].numpy() == cls)
            assert (np.isnan(d_h) and np.isnan(d_r)) or abs(d_h - d_r) < 1e-4
    s.running_metric.reset()
    s.evaluate(dev(F_["vol"]), F_["vlab"].numpy(), n_iter=2)
    ref_hist = sum(O.confusion_hist(F_["vlab"][i].numpy(), ops.argmax_c(p2).cpu()[i].numpy().astype(np.int64), 4) for i in range(3))
    assert np.array_equal(s.running_metric.confusion_matrix, ref_hist)


def test_bs16_256_forward_checksum_vs_golden(golden_cases, golden_sd):
    """BASELINE-sized input (16 x 256 x 256): the four standard-training losses and the latent codes' checksums."""
    G = golden_cases["G_bs16_256_fwd"]
    s = _solver(golden_sd)
    c, l, n = O.synthetic_batch(16, 256, 256, seed=0)
    s.train()
    with torch.no_grad():
        st = s.standard_training(dev(c), dev(l), dev(n))
    got = torch.stack([v.detach().float() for v in st]).cpu().double()
    assert torch.allclose(got, G["losses"], atol=1e-4), (got, G["losses"])
    for z, key in ((s.z_i, "z_i_stats"), (s.z_s, "z_s_stats")):
        zz = z.detach().double().cpu()
        stt = torch.tensor([zz.sum().item(), zz.norm().item(), zz.abs().max().item()], dtype=torch.float64)
        assert torch.allclose(stt, G[key], rtol=2e-4), (stt, G[key])


def test_full_size_step_properties(golden_sd):
    """At the metric's size (bs16, 256x256) run one full step and check size-independent properties: finite losses,
    exactly k masked entries per image, masked code == code * mask, BN running stats moved, weights changed by <= lr."""
    s = _solver(golden_sd)
    c, l, n = O.synthetic_batch(16, 256, 256, seed=3)
    img_cfg = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    seg_cfg = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    before = s.model["shape_decoder"]._flat_data.clone()
    losses = s.cooperative_step(dev(c), dev(l), dev(n), img_cfg, seg_cfg)
    vals = torch.stack([v.detach().float() for v in losses]).cpu()
    assert torch.isfinite(vals).all() and (vals >= 0).all()
    mi, ms = s.last_masks["image"].cpu(), s.last_masks["seg"].cpu()
    assert mi.shape == (16, 128, 1, 1) and ms.shape == (16, 1, 16, 16)
    assert ((mi == 0).flatten(1).sum(1) == 64).all() and ((ms == 0).flatten(1).sum(1) == 128).all()
    delta = (s.model["shape_decoder"]._flat_data - before).abs().max().item()
    assert 0 < delta <= 1.001e-4           # |Adam's first step| <= lr (+ fp32 rounding of the weight)
    assert int(s.model["image_encoder"]._nbt[0]) == 1 and int(s.model["image_decoder"]._nbt[0]) == 3


def test_bench_two_ranks_control_flow():
    """bench.py under the driver's launcher with TWO ranks (both on GPU 0, gloo): every rank must reach every collective (timed
    steps, barrier, the single-stream replay after the timed region) -- a rank-0-only step would hang here."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64",
           "--batch", "4", "--backend", "gloo", "--all-on-device0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]          # the driver parses the LAST line of stdout
    assert len(line) < 3072, len(line)
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["value"] > 0


# ---------------------------------------------------------------------------------------------- SURVEY 8(f) rows 1, 2, 4
def test_checkpoint_interop_and_resume(golden_cases, golden_sd, tmp_path):
    """save_model writes upstream-format state dicts (model.py:666-678): they load strictly into the oracle's torch modules
    (= the reference's key names and shapes) and into a fresh engine; a snapshot (model.py:680-738) resumes training bit-exactly."""
    C = golden_cases["D_step_dropout"]
    ov = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    args = (dev(C["clean"]), dev(C["label"]), dev(C["noisy"]), C["img_cfg"], C["seg_cfg"])
    a = _solver(golden_sd)
    a.cooperative_step(*args, image_override=ov[0], seg_override=ov[1])
    a.save_model(str(tmp_path), 3, save_optimizers=True)
    snap = a.save_snapshots(str(tmp_path), epoch=3)
    ckpt = tmp_path / "3" / "checkpoints"
    onets = O.build_networks(init=False)
    for name, net in onets.items():
        missing = net.load_state_dict(torch.load(ckpt / f"{name}.pth"), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    b = AdvancedTripletReconSegmentationModel(use_gpu=True, checkpoint_dir=str(ckpt))
    x = dev(C["noisy"])
    assert torch.equal(a.predict(x), b.predict(x))
    c = _solver(golden_sd)
    assert c.load_snapshots(snap) == 3
    la = a.cooperative_step(*args, image_override=ov[0], seg_override=ov[1])
    lc = c.cooperative_step(*args, image_override=ov[0], seg_override=ov[1])
    assert all(torch.equal(u.detach(), v.detach()) for u, v in zip(la, lc))
    for k in a.model:
        assert torch.equal(a.model[k]._flat_data, c.model[k]._flat_data), k


@pytest.mark.parametrize("variant", ["FCN_16_standard_share_code", "FCN_16_standard_w_o_filter"])
def test_ablation_variants_vs_oracle(golden_cases, golden_sd, variant):
    """model.py:199-203: share_code (z_i := z_s) and w_o_filter (z_s := z_i) are flags of the same solver."""
    A = golden_cases["A_standard"]
    s = AdvancedTripletReconSegmentationModel(network_type=variant, use_gpu=True)
    for k, m in s.model.items():
        m.load_state_dict(golden_sd[k])
    o = O.OracleSolver(state_dicts=golden_sd, network_type=variant)
    got = s.standard_training(dev(A["clean"]), dev(A["label"]), dev(A["noisy"]))
    ref = o.standard_training(A["clean"], A["label"], A["noisy"])
    for g, r in zip(got, ref):
        assert abs(float(g) - float(r)) < 1e-4
    close(s.z_i, o.z_i, what="z_i")
    close(s.z_s, o.z_s, what="z_s")
    assert torch.equal(s.z_i, s.z_s)


def test_evaluate_with_device_targets_matches_host_path(golden_cases, golden_sd):
    F_ = golden_cases["F_predict"]
    a, b = _solver(golden_sd), _solver(golden_sd)
    a.evaluate(dev(F_["vol"]), F_["vlab"].numpy(), n_iter=2)              # upstream path: numpy targets, host confusion matrix
    b.evaluate(dev(F_["vol"]), dev(F_["vlab"]), n_iter=2)                  # device targets: confusion matrix stays on the GPU
    sa, ia = a.running_metric.get_scores()
    sb, ib = b.running_metric.get_scores()
    assert sa == sb and all((ia[k] == ib[k]) or (np.isnan(ia[k]) and np.isnan(ib[k])) for k in ia)


class _VolumeSet:
    """The slice of the reference dataset interface the patient-wise tester reads (test_basic_segmentation_solver.py:68-72, 98)."""
    formalized_label_dict = {0: "BG", 1: "LV", 2: "MYO", 3: "RV"}

    def __init__(self, volumes):
        self.volumes, self.patient_number, self._cur = volumes, len(volumes), None

    def get_patient_data_for_testing(self, i, crop_size=None):
        self._cur = i
        return {"image": self.volumes[i][0], "label": self.volumes[i][1]}

    def get_id(self):
        return "patient%03d" % self._cur

    def get_voxel_spacing(self):
        return [1.25, 1.25, 10.0]


def test_patient_wise_tester_volume_in_dice_out(golden_cases, golden_sd, tmp_path):
    """SURVEY 8(f) row 1 end to end: chunked predict + device arg-max + device voxel counts give the rows the reference's
    mask-by-mask bookkeeping gives on the same label maps (exactly), and those label maps are the reference's own away from ties."""
    from cooperative_training_and_latent_space_data_augmentation_amd.tester import TestSegmentationNetwork
    F_ = golden_cases["F_predict"]
    s = _solver(golden_sd)
    s.train()
    with torch.no_grad():
        for i in range(3):
            c_, l_, n_ = O.synthetic_batch(2, 64, 64, seed=10 + i, structured=True)
            s.standard_training(dev(c_), dev(l_), dev(n_))
    s.n_iter = 2                                                            # FTN + STN refinement (model.py:375-394)
    vol, lab = F_["vol"], F_["vlab"]
    data = _VolumeSet([(vol, lab), (torch.cat([vol, vol.flip(0)], 0), torch.cat([lab, lab.flip(0)], 0))])
    t = TestSegmentationNetwork(data, crop_size=None, segmentation_model=s, save_path=str(tmp_path), metrics_list=["Dice", "VolError"])
    pack = data.get_patient_data_for_testing(0)
    pid, res = t.evaluate(0, pack, 2, maximum_batch_size=2)                 # chunks of 2 + 1 slices
    top2 = F_["logits_n2"].topk(2, dim=1)[0]
    safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
    assert pid == "patient000" and np.array_equal(res["pred"][safe], F_["argmax_n2"].numpy()[safe])
    assert np.abs(res["soft_pred"] - F_["logits_n2"].numpy()).max() < 2e-4
    row = t.segmentation_metric.tables[0]
    assert row[1:] == O.patient_scores(res["pred"], lab.numpy(), data.formalized_label_dict.keys(), metrics=("Dice", "VolError"))
    ref = O.patient_scores(F_["argmax_n2"].numpy(), lab.numpy(), data.formalized_label_dict.keys(), metrics=("Dice", "VolError"))
    assert max(abs(a - b) for a, b in zip(row[1:], ref)) < 1e-3
    t.segmentation_metric.reset()
    df = t.run()
    assert list(df.columns) == ["patient_id", "LV_Dice", "LV_VolError", "MYO_Dice", "MYO_VolError", "RV_Dice", "RV_VolError"]
    assert list(df["patient_id"]) == ["patient000", "patient001"] and (tmp_path / "result.csv").exists() and (tmp_path / "details.csv").exists()
    # a volume and the same volume followed by its mirror in the slice order have the same per-class counts ratio -> the same Dice
    assert np.allclose(df.iloc[0, 1::2].to_numpy(float), df.iloc[1, 1::2].to_numpy(float), atol=1e-12)
    assert t.get_top_k_results(1, "MYO_Dice").shape[0] == 1


def test_whole_volume_pass_is_bitwise_the_chunked_loop(golden_sd):
    """`predict` uses eval-mode BatchNorm, so slices are independent and a 40-slice 192x192 volume run
    as ONE pass must give exactly the reference loop's <= 10-slice chunks (test_basic_segmentation_solver.py:85-114): logits and the
    uint8 label volume bit for bit, for n_iter 1 and 2; plus the 32-bit-offset guard of the pass size."""
    from cooperative_training_and_latent_space_data_augmentation_amd.tester import max_slices_per_pass, predict_volume
    s = _solver(golden_sd)
    s.train()
    with torch.no_grad():                                   # non-trivial running statistics
        for i in range(2):
            c_, l_, n_ = O.synthetic_batch(2, 64, 64, seed=20 + i, structured=True)
            s.standard_training(dev(c_), dev(l_), dev(n_))
    vol = dev(torch.rand(40, 1, 192, 192, generator=torch.Generator().manual_seed(9)))
    for n_iter in (1, 2):
        whole = s.predict(vol, n_iter=n_iter)
        parts = torch.cat([s.predict(vol[lo:lo + 10], n_iter=n_iter) for lo in range(0, 40, 10)], 0)
        assert torch.equal(whole, parts)
        # the reference's literal loop (coalesce=False), the default call with the reference's arguments (its chunks run as one pass) and
        # the explicit whole-volume call: the same labels
        literal = predict_volume(s, vol, n_iter=n_iter, chunk=10, coalesce=False)
        n0 = _ffi.lib.ctl_launch_count()
        default = predict_volume(s, vol, n_iter=n_iter, chunk=10)
        n1 = _ffi.lib.ctl_launch_count()
        predict_volume(s, vol, n_iter=n_iter, chunk=10, coalesce=False)
        n2 = _ffi.lib.ctl_launch_count()
        assert torch.equal(predict_volume(s, vol, n_iter=n_iter, chunk=None), literal) and torch.equal(default, literal)
        assert (n2 - n1) > 3 * (n1 - n0)                                   # ... in a quarter of the launches
        assert torch.equal(predict_volume(s, vol, n_iter=n_iter, chunk=7, coalesce=False), literal)     # ragged tail
    assert max_slices_per_pass(192, 192) == (2 ** 31 - 1) // (192 * 192 * 64) == 910
    assert max_slices_per_pass(4096, 4096) == 1
    # ... and by half of the free device memory at 512 floats per pixel and slice (`maximum_batch_size` bounds memory upstream)
    free, _ = torch.cuda.mem_get_info(vol.device)
    assert max_slices_per_pass(192, 192, device=vol.device) == max(1, min(910, (free // 2) // (192 * 192 * 4 * 512)))
    assert max_slices_per_pass(2048, 2048, device=vol.device) <= max(1, (free // 2) // (2048 * 2048 * 4 * 512))


@pytest.mark.parametrize("two_streams", [True, False])
@pytest.mark.parametrize("case", ["C_step_channel_spatial", "E_step_soft_random"])
def test_saliency_forward_reuse_is_bitwise(golden_cases, golden_sd, case, two_streams):
    """Targeted masks: the saliency pass decodes, in training mode, the code the standard pass has just decoded (model_util.py:214 after
    model.py:436-447).  The engine re-uses the standard pass' activations (CtlNet.reuse_pass), runs only the data-gradient backward on
    them and replays the second running-statistics update from the saved batch statistics (ctl_bn_replay_running).  Three training steps
    with and without the re-use must agree bit for bit: losses, masks, weights, BatchNorm buffers -- and the re-use must save launches."""
    from cooperative_training_and_latent_space_data_augmentation_amd import model_util as MU
    C = golden_cases[case]
    res = []
    for reuse in (False, True):
        MU.REUSE_SALIENCY_FORWARD = reuse
        try:
            s = _solver(golden_sd)
            s.two_streams = two_streams
            ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
            n0 = _ffi.lib.ctl_launch_count()
            for _ in range(3):
                losses = s.cooperative_step(dev(C["clean"]), dev(C["label"]), dev(C["noisy"]), C["img_cfg"], C["seg_cfg"], image_override=ov_img,
                                            seg_override=ov_seg)
            torch.cuda.synchronize()
            res.append((torch.stack([v.detach().float() for v in losses]).cpu(), {k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()},
                        {k: (m._bflat.detach().cpu().clone(), m._nbt.detach().cpu().clone()) for k, m in s.model.items()},
                        {k: v.detach().cpu().clone() for k, v in s.last_masks.items()}, int(_ffi.lib.ctl_launch_count() - n0)))
        finally:
            MU.REUSE_SALIENCY_FORWARD = True
    a, b = res
    assert torch.equal(a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
        assert torch.equal(a[2][k][0], b[2][k][0]) and torch.equal(a[2][k][1], b[2][k][1]), f"BatchNorm buffers of {k}"
    assert all(torch.equal(a[3][k], b[3][k]) for k in a[3])
    assert b[4] <= a[4] - 3 * 2 * 20, (a[4], b[4])              # two decoder forwards (~25 launches each, minus the replay launch) less per step


def test_filter_code_and_remaining_solver_entries(golden_cases, golden_sd):
    """`Dual_Branch_Encoder.filter_code` (encoder_decoder.py:496-498) and the solver entries built on it
    (model.py:208-221, 292-295, 603-606): same numbers as the full forward / the oracle, forward only."""
    onet, hnet = make_pair("image_encoder", golden_sd)
    x = torch.rand(2, 1, 64, 48, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        for mode in ("train", "eval"):          # train: batch statistics, running statistics move once per call on both sides
            if mode == "eval":
                onet.eval(); hnet.eval()
            z_i, z_s = hnet(dev(x))
            f = hnet.filter_code(z_i)
            assert torch.equal(f, z_s)          # same kernels on the same z_i: bitwise
            close(f, onet.filter_code(onet(x)[0]), what=f"filter_code ({mode} mode)")
    with pytest.raises(_ffi.CtlError):
        hnet.filter_code(z_i.clone().requires_grad_(True))
    with pytest.raises(ValueError):
        hnet.filter_code(torch.zeros(1, 16, 4, 4, device=DEV))
    s = _solver(golden_sd)
    s.eval()
    vol = dev(golden_cases["F_predict"]["vol"])
    with torch.no_grad():
        (z_i, z_s), pred = s.fast_predict(vol)
        assert torch.equal(s.decode_segmentation_from_image_code(z_i), pred)
        assert torch.equal(s.predict_w_reconstructed_image(vol), s.fast_predict(s.recon_image(vol))[1])
    s.requires_grad_(False)
    assert not any(p.requires_grad for m in s.model.values() for p in m.parameters())
    s.requires_grad_(True)
    assert all(p.requires_grad for m in s.model.values() for p in m.parameters())


def test_adam_skips_a_network_without_gradient(golden_cases, golden_sd):
    """torch.optim.Adam skips parameters whose .grad is None (zero_grad sets grads to None since torch 2.0): a network that got no
    gradient in a step must not move, even with non-zero Adam moments; a network that did get one moves."""
    C = golden_cases["D_step_dropout"]
    ov = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    s = _solver(golden_sd)
    s.cooperative_step(dev(C["clean"]), dev(C["label"]), dev(C["noisy"]), C["img_cfg"], C["seg_cfg"], image_override=ov[0], seg_override=ov[1])
    before = {k: m._flat_data.clone() for k, m in s.model.items()}
    s.reset_all_optimizers()
    rec = s.recon_shape(dev(C["label"]), is_label_map=True)                 # STN only: three networks never see a gradient
    from cooperative_training_and_latent_space_data_augmentation_amd.autograd import cross_entropy_2D
    cross_entropy_2D(rec, dev(C["label"])).backward()
    s.optimize_all_params()
    for k in ("image_encoder", "segmentation_decoder", "image_decoder"):
        assert torch.equal(s.model[k]._flat_data, before[k]) and s.optimizers[k].step_count == 1, k
    for k in ("shape_encoder", "shape_decoder"):
        assert not torch.equal(s.model[k]._flat_data, before[k]) and s.optimizers[k].step_count == 2, k


def test_host_side_caches_keep_module_semantics(golden_sd):
    """CtlNet.param_list / train() fast path / the cached BatchNorm parameter list (host time, not results): requires_grad and training
    flags still behave like nn.Module's."""
    from cooperative_training_and_latent_space_data_augmentation_amd.model_util import set_grad
    net = nets.build_networks(device=DEV, state_dicts={"shape_decoder": golden_sd["shape_decoder"]})["shape_decoder"]
    assert [id(p) for p in net.param_list()] == [id(p) for p in net.parameters()] and net.param_list() is net.param_list()
    set_grad(net, False)
    assert not net.wants_param_grad() and not any(p.requires_grad for p in net.parameters())
    next(iter(net.parameters())).requires_grad = True               # a single parameter flipped by hand is seen
    assert net.wants_param_grad()
    set_grad(net, True)
    assert all(p.requires_grad for p in net.parameters())
    net.eval()
    assert not net.training and not any(m.training for m in net.modules())
    net.train()
    assert net.training and all(m.training for m in net.modules())
    next(net.children()).eval()                                     # a child flipped by hand: the next train() repairs the tree
    net.train(False)
    net.train(True)
    assert all(m.training for m in net.modules())
    with _disable_tracking_bn_stats(net):
        assert net._bn_track is False and not any(p.requires_grad for p in net.__dict__["_bn_plist"])
    assert all(p.requires_grad for p in net.__dict__["_bn_plist"])


def test_spin_kernel_and_sampled_profiler():
    """ctl_spin (the idle kernel of the stream-pair probe) waits about as long as asked; the in-process profiler brackets every n-th launch."""
    from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check
    st = torch.cuda.current_stream().cuda_stream
    check(lib.ctl_spin(1, st), "ctl_spin")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.ctl_spin(500, st), "ctl_spin")
    e1.record()
    e1.synchronize()
    assert 450.0 <= e0.elapsed_time(e1) * 1e3 <= 900.0, e0.elapsed_time(e1)
    x = dev(torch.randn(2, 16, 32, 32))
    w = ops.pack_oihw_fwd(dev(torch.randn(16, 16, 3, 3) * 0.1))
    d = _ffi.conv_desc(n=2, hin=32, win=32, cin=16, hout=32, wout=32, cout=16, ks=3)
    for every, want in ((1, 10), (4, 3)):
        _ffi.prof_start("conv_igemm", every)
        for _ in range(10):
            ops.conv_forward(d, x, w)
        rec = _ffi.prof_stop()
        assert len(rec) == 1 and int(next(iter(rec.values()))["launches"]) == want, (every, rec)


@pytest.mark.parametrize("dtype,masks,budget", [("fp32", "dropout", 860), ("bf16", "targeted", 925)])
def test_step_launch_budget(dtype, masks, budget):
    """The library launches of one cooperative step at the bench size -- a guard for the fusions of round 3 (the backward of a residual block
    has no element-wise BatchNorm pass left: reductions ride in the epilogues of the producing convs, applies in the staging of the consuming
    ones, sum-pools in the tail epilogue).  End of round 2: 1035 (fp32, dropout masks) / 1120 (bf16, targeted masks); now 845 / 906."""
    import bench
    s = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype=dtype)
    clean, label, noisy, _ = bench.synthetic(16, 256, 256, 7, torch.device(DEV))
    cfg = (bench.TGT_IMG, bench.TGT_SEG) if masks == "targeted" else (bench.DROP_IMG, bench.DROP_SEG)
    for _ in range(2):
        s.cooperative_step(clean, label, noisy, *cfg)
    torch.cuda.synchronize()
    n0 = _ffi.lib.ctl_launch_count()
    s.cooperative_step(clean, label, noisy, *cfg)
    torch.cuda.synchronize()
    n = int(_ffi.lib.ctl_launch_count() - n0)
    print(f"{dtype} {masks}: {n} library launches per step")
    assert n <= budget, (n, budget)
