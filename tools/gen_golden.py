#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only at /root/reference) on CPU.

Run in the build container only:  python tools/gen_golden.py
The reference never travels: only inputs/expected outputs (plain tensors) are written to tests/golden/.
Missing third-party imports of the reference (SimpleITK, medpy, ...) are stubbed exactly as SURVEY.md 8(c) lists.
"""
import os
import sys
from unittest.mock import MagicMock

for _m in ["numpy.lib.function_base", "SimpleITK", "medpy", "medpy.metric", "medpy.metric.binary", "IPython",
           "IPython.display", "skimage", "skimage.transform", "seaborn"]:
    sys.modules[_m] = MagicMock()
REF = os.environ.get("CTL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from medseg.models.advanced_triplet_recon_segmentation_model import AdvancedTripletReconSegmentationModel  # noqa: E402
import medseg.models.model_util as ref_mu  # noqa: E402
import medseg.models.advanced_triplet_recon_segmentation_model as ref_model  # noqa: E402
from oracle.ref_cpu import synthetic_batch, NET_NAMES  # noqa: E402  (inputs only; nothing of the oracle's math is used)


def new_solver(seed=0, network_type="FCN_16_standard"):
    torch.manual_seed(seed)
    return AdvancedTripletReconSegmentationModel(network_type=network_type, image_ch=1, num_classes=4,
                                                 learning_rate=1e-4, n_iter=1, use_gpu=False)


def clone_sd(solver):
    return {k: {n: t.detach().clone() for n, t in m.state_dict().items()} for k, m in solver.model.items()}


def tensor_stats(t):
    t = t.detach().double()
    return torch.tensor([t.sum().item(), t.norm().item(), t.abs().max().item()], dtype=torch.float64)


def grad_stats(solver):
    out = {}
    for k, m in solver.model.items():
        for n, p in m.named_parameters():
            out[f"{k}/{n}"] = None if p.grad is None else tensor_stats(p.grad)
    return out


def param_stats(solver):
    return {f"{k}/{n}": tensor_stats(p) for k, m in solver.model.items() for n, p in m.named_parameters()}


def buffer_dump(solver):
    return {f"{k}/{n}": b.detach().clone() for k, m in solver.model.items() for n, b in m.named_buffers()}


PICK_GRADS = ["image_encoder/general_encoder.inc.0.weight", "image_encoder/general_encoder.down1.conv.3.weight",
              "image_encoder/general_encoder.down4.conv_input.weight", "image_encoder/code_decoupler.0.bias",
              "image_encoder/general_encoder.final_conv.1.weight",
              "segmentation_decoder/up4.conv.0.weight", "segmentation_decoder/final_conv.weight",
              "segmentation_decoder/up1.conv.1.bias",
              "shape_encoder/inc.0.weight", "shape_encoder/down2.down.weight", "shape_encoder/down3.conv.4.weight",
              "shape_decoder/up3.conv_input.weight", "shape_decoder/final_conv.bias",
              "image_decoder/up4.up.weight", "image_decoder/up2.up.bias", "image_decoder/final_conv.weight"]


def pick(solver, what):
    out = {}
    for key in PICK_GRADS:
        k, n = key.split("/")
        p = dict(solver.model[k].named_parameters())[n]
        t = p.grad if what == "grad" else p
        out[key] = None if t is None else t.detach().clone()      # (ablation variants leave whole branches without a gradient)
    return out


CFG_CH_MSE = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
CFG_SP_CE = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
CFG_SP_MSE = {"loss_name": "mse", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
CFG_CH_CE = {"loss_name": "ce", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
CFG_DROP_MSE = {"loss_name": "mse", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
CFG_DROP_CE = {"loss_name": "ce", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}


def ref_step(solver, clean, label, noisy, img_cfg, seg_cfg, record, separate_training=False, keep_big=True):
    """train_adv_supervised_segmentation_triplet.py:171-231 with the noise supplied; records every random draw.
    keep_big=False (metric-sized cases): checksums instead of the activation-sized tensors."""
    solver.train()
    solver.reset_all_optimizers()
    std = solver.standard_training(clean, label, perturbed_image=noisy, separate_training=separate_training)
    standard_loss = std[0] + std[1] + std[3] + std[2]
    if keep_big:
        record["z_i"], record["z_s"] = solver.z_i.detach().clone(), solver.z_s.detach().clone()
    record["z_i_stats"], record["z_s_stats"], record["z_shape"] = tensor_stats(solver.z_i), tensor_stats(solver.z_s), tuple(solver.z_i.shape)
    solver.reset_all_optimizers()

    # capture masks / k / soft-noise drawn inside the reference
    masks, ks, noises, keeps = [], [], [], []
    orig_perturb = solver.perturb_latent_code
    orig_rand, orig_rand_like = np.random.rand, torch.rand_like
    orig_dropout2d = ref_model.F.dropout2d

    def dropout2d(inp, p=0.5, *a, **kw):
        out = orig_dropout2d(inp, p, *a, **kw)
        # keep[n,c]=1 if the channel survived (an all-zero input channel is indistinguishable and irrelevant)
        keeps.append(((out != 0).flatten(2).any(2) | (inp == 0).flatten(2).all(2)).float())
        return out

    def perturb(*a, **kw):
        z, m = orig_perturb(*a, **kw)
        masks.append(m.detach().clone())
        return z, m

    def rand(*a):
        v = orig_rand(*a)
        ks.append(float(v))
        return v

    def rand_like(t, *a, **kw):
        v = orig_rand_like(t, *a, **kw)
        noises.append(v.detach().clone())
        return v

    solver.perturb_latent_code = perturb
    ref_mu.np.random.rand = rand
    torch.rand_like = rand_like
    ref_model.F.dropout2d = dropout2d
    try:
        xh, yh = solver.hard_example_generation(clean.detach().clone(), label.detach().clone(),
                                                gen_corrupted_seg=True, gen_corrupted_image=True,
                                                corrupted_image_DA_config=img_cfg, corrupted_seg_DA_config=seg_cfg)
    finally:
        solver.perturb_latent_code = orig_perturb
        ref_mu.np.random.rand = orig_rand
        torch.rand_like = orig_rand_like
        ref_model.F.dropout2d = orig_dropout2d
    record["masks"], record["rand_draws"], record["soft_noises"], record["dropout_keeps"] = masks, ks, noises, keeps
    if keep_big:
        record["x_hard"], record["y_hard"] = xh.detach().clone(), yh.detach().clone()
    else:       # dropout masks of the reference are activation-sized equality masks: keep the [N,C] / [N,L] form only
        record["masks"] = [m if m.numel() <= 1 << 16 else None for m in masks]
    record["x_hard_stats"], record["y_hard_stats"] = tensor_stats(xh), tensor_stats(yh)
    hard = solver.hard_example_training(perturbed_image=xh, perturbed_seg=yh, clean_image_l=clean, label_l=label,
                                        separate_training=separate_training, use_gpu=False)
    hard_loss = hard[0] + hard[1] + hard[2] + hard[3]
    loss = standard_loss + hard_loss
    solver.reset_all_optimizers()
    loss.backward()
    record["losses"] = torch.tensor([float(v) for v in std] + [float(v) for v in hard], dtype=torch.float64)
    record["grad_stats"] = grad_stats(solver)
    record["grads"] = pick(solver, "grad")
    solver.optimize_all_params()
    record["param_stats_after"] = param_stats(solver)
    record["params_after"] = pick(solver, "param")
    record["buffers_after"] = buffer_dump(solver)


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}

    # ---- initial weights, seed 0
    s = new_solver(0)
    sd0 = clone_sd(s)
    torch.save(sd0, os.path.join(OUT, "state_dicts_seed0.pt"))

    # ---- case A: standard_training forward + backward (N=2, 64x64)
    clean, label, noisy = synthetic_batch(2, 64, 64, seed=1, structured=True)
    s.train()
    s.reset_all_optimizers()
    std = s.standard_training(clean, label, perturbed_image=noisy)
    (std[0] + std[1] + std[2] + std[3]).backward()
    A = {"clean": clean, "label": label, "noisy": noisy,
         "losses": torch.tensor([float(v) for v in std], dtype=torch.float64),
         "z_i": s.z_i.detach().clone(), "z_s": s.z_s.detach().clone(),
         "grad_stats": grad_stats(s), "grads": pick(s, "grad"), "buffers_after": buffer_dump(s)}
    cases["A_standard"] = A

    # ---- case B: the two masking functions on the codes of case A (fresh solver => same weights; the
    # saliency forward runs the decoder in train mode and therefore moves its running stats: recorded too)
    s = new_solver(0)
    s.train()
    B = {}
    z_i, z_s = A["z_i"], A["z_s"]
    from medseg.common_utils.basic_operations import set_grad
    set_grad(s.model["segmentation_decoder"], False)
    set_grad(s.model["image_decoder"], False)
    for name, fn, z, dec, lab, loss_type, ncls in [
        ("channel_mse", ref_mu.mask_latent_code_channel_wise, z_i, "image_decoder", clean, "mse", 4),
        ("spatial_mse", ref_mu.mask_latent_code_spatial_wise, z_i, "image_decoder", clean, "mse", 4),
        ("channel_ce", ref_mu.mask_latent_code_channel_wise, z_s, "segmentation_decoder", label, "ce", 4),
        ("spatial_ce", ref_mu.mask_latent_code_spatial_wise, z_s, "segmentation_decoder", label, "ce", 4),
    ]:
        for pct in (0.5, 0.2):
            masked, mask = fn(z, num_classes=ncls, decoder_function=s.model[dec], label=lab, percentile=pct,
                              random=False, loss_type=loss_type, if_detach=True, if_soft=False)
            B[f"{name}_p{pct}"] = {"masked": masked.detach().clone(), "mask": mask.detach().clone()}
        torch.manual_seed(77)
        masked, mask = fn(z, num_classes=ncls, decoder_function=s.model[dec], label=lab, percentile=0.3,
                          random=False, loss_type=loss_type, if_detach=True, if_soft=True)
        B[f"{name}_soft_seed77"] = {"masked": masked.detach().clone(), "mask": mask.detach().clone()}
    B["buffers_after"] = {k: v for k, v in buffer_dump(s).items() if k.split("/")[0] in ("image_decoder", "segmentation_decoder")}
    cases["B_masking"] = B

    # ---- case C: full cooperative step, targeted masks, deterministic (hard masks, fixed threshold)
    s = new_solver(0)
    C = {"clean": clean, "label": label, "noisy": noisy, "img_cfg": CFG_CH_MSE, "seg_cfg": CFG_SP_CE}
    ref_step(s, clean, label, noisy, CFG_CH_MSE, CFG_SP_CE, C)
    cases["C_step_channel_spatial"] = C

    # ---- case D: full step, dropout on both codes (config 2 of BASELINE.json); masks recorded
    s = new_solver(0)
    torch.manual_seed(5)
    D = {"clean": clean, "label": label, "noisy": noisy, "img_cfg": CFG_DROP_MSE, "seg_cfg": CFG_DROP_CE}
    ref_step(s, clean, label, noisy, CFG_DROP_MSE, CFG_DROP_CE, D)
    cases["D_step_dropout"] = D

    # ---- case E: full step, spatial(mse)+channel(ce), random threshold + soft masks; draws recorded
    s = new_solver(0)
    torch.manual_seed(6)
    np.random.seed(6)
    clean3, label3, noisy3 = synthetic_batch(3, 48, 48, seed=2, structured=False)
    E = {"clean": clean3, "label": label3, "noisy": noisy3, "img_cfg": CFG_SP_MSE, "seg_cfg": CFG_CH_CE}
    ref_step(s, clean3, label3, noisy3, CFG_SP_MSE, CFG_CH_CE, E)
    cases["E_step_soft_random"] = E

    # ---- case F: inference after 3 train-mode forward passes (moves the running stats, no optimizer step)
    s = new_solver(0)
    s.train()
    with torch.no_grad():
        for i in range(3):
            c_, l_, n_ = synthetic_batch(2, 64, 64, seed=10 + i, structured=True)
            s.standard_training(c_, l_, perturbed_image=n_)
    vol, vlab, _ = synthetic_batch(3, 48, 48, seed=20, structured=True)
    p1 = s.predict(vol, n_iter=1).detach().clone()
    p2 = s.predict(vol, n_iter=2).detach().clone()
    cases["F_predict"] = {"vol": vol, "vlab": vlab, "logits_n1": p1, "logits_n2": p2,
                          "argmax_n1": p1.max(1)[1].to(torch.uint8), "argmax_n2": p2.max(1)[1].to(torch.uint8),
                          "buffers_after": buffer_dump(s)}

    # ---- case G: run-to-run determinism witness + bs16/256 checksum (metric-sized, checksum only)
    s = new_solver(0)
    c16, l16, n16 = synthetic_batch(16, 256, 256, seed=0, structured=False)
    s.train()
    with torch.no_grad():
        st = s.standard_training(c16, l16, perturbed_image=n16)
    cases["G_bs16_256_fwd"] = {"losses": torch.tensor([float(v) for v in st], dtype=torch.float64),
                               "z_i_stats": tensor_stats(s.z_i), "z_s_stats": tensor_stats(s.z_s)}

    torch.save(cases, os.path.join(OUT, "cases.pt"))
    for k in cases:
        print("wrote case", k)
    print("sizes:", {f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)})


if __name__ == "__main__":
    main()
