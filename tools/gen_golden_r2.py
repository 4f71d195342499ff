#!/usr/bin/env python3
"""Round-2 golden vectors, recorded from the REAL reference (read-only at /root/reference) on CPU:  python tools/gen_golden_r2.py
-> tests/golden/cases_r2.pt.  Every BASELINE.json fp32 config at its real size plus the option surface of SURVEY 8(f) row 4:

  H_bs16_dropout_step      config 2: bs16 x 256^2 full cooperative step, dropout on both codes (keep patterns recorded)
  I_bs16_targeted_step     config 3 (fp32 arithmetic): bs16 x 256^2, channel(mse) + spatial(ce), random threshold + soft masks
  J_predict_192            config 5: one 10 x 1 x 192 x 192 chunk, eval BatchNorm, n_iter = 1 / 2 / 3 (logit samples + full label maps)
  K_separate_training      full step with separate_training=True (model.py:458-462, 552-553)
  L_share_code / M_w_o_filter   full step (with backward) of the ablation variants (model.py:199-203)
  R_random_scheme          mask_type='random' + random_threshold from SEEDED python `random` / `np.random`: the scheme and k sequence of
                           six generation calls and one full step (config 4's masking)

Metric-sized cases keep checksums (sum, L2, max|.|) instead of activation-sized tensors; inputs are regenerated from seeds by
`oracle.ref_cpu.synthetic_batch` on the test side.  Only data is written -- the reference never travels."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                      # noqa: E402  (stubs the missing third-party imports and imports the reference)
import numpy as np                          # noqa: E402
import torch                                # noqa: E402

from oracle.ref_cpu import synthetic_batch  # noqa: E402  (inputs only)

CFG_CH_MSE_SOFT = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
CFG_SP_CE_SOFT = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
CFG_RAND_MSE = {"loss_name": "mse", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": False}
CFG_RAND_CE = {"loss_name": "ce", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": False}


def scheme_of(mask, code_shape):
    n, c, h, w = code_shape
    if tuple(mask.shape) == (n, c, 1, 1):
        return "channel"
    if tuple(mask.shape) == (n, 1, h, w):
        return "spatial"
    return "dropout"


def move_running_stats(s, n, size, passes=3, seed0=10):
    s.train()
    with torch.no_grad():
        for i in range(passes):
            c_, l_, n_ = synthetic_batch(n, size, size, seed=seed0 + i, structured=True)
            s.standard_training(c_, l_, perturbed_image=n_)


def main():
    cases = {}
    # ---- H / I: metric-sized full steps (BASELINE configs 2 and 3)
    for name, seed, tseed, cfgs in (("H_bs16_dropout_step", 0, 15, (G.CFG_DROP_MSE, G.CFG_DROP_CE)),
                                    ("I_bs16_targeted_step", 3, 16, (CFG_CH_MSE_SOFT, CFG_SP_CE_SOFT))):
        s = G.new_solver(0)
        torch.manual_seed(tseed)
        np.random.seed(tseed)
        clean, label, noisy = synthetic_batch(16, 256, 256, seed=seed)
        rec = {"batch": (16, 256, 256, seed), "img_cfg": cfgs[0], "seg_cfg": cfgs[1]}
        G.ref_step(s, clean, label, noisy, cfgs[0], cfgs[1], rec, keep_big=False)
        rec["buffer_stats_after"] = {k: G.tensor_stats(v) for k, v in rec["buffers_after"].items()}
        cases[name] = rec
        print("wrote", name, rec["losses"].tolist(), flush=True)

    # ---- J: config 5 at its real shape
    s = G.new_solver(0)
    move_running_stats(s, 4, 192)
    vol, vlab, _ = synthetic_batch(10, 192, 192, seed=21, structured=True)
    J = {"batch": (10, 192, 192, 21), "buffers_after": G.buffer_dump(s)}
    for it in (1, 2, 3):
        p = s.predict(vol, n_iter=it).detach()
        top2 = p.topk(2, dim=1)[0]
        J[f"logit_stats_n{it}"] = G.tensor_stats(p)
        J[f"logits_sub_n{it}"] = p[:, :, ::8, ::8].clone()                 # every 8th pixel: 10 x 4 x 24 x 24
        J[f"argmax_n{it}"] = p.max(1)[1].to(torch.uint8)
        J[f"safe_n{it}"] = (top2[:, 0] - top2[:, 1]) > 1e-3                # label maps are compared bit-exactly away from near-ties
    cases["J_predict_192"] = J
    print("wrote J_predict_192", flush=True)

    # ---- K / L / M: option surface, small shapes, deterministic masks
    clean, label, noisy = synthetic_batch(2, 64, 64, seed=1, structured=True)
    for name, ntype, sep in (("K_separate_training", "FCN_16_standard", True),
                             ("L_share_code", "FCN_16_standard_share_code", False),
                             ("M_w_o_filter", "FCN_16_standard_w_o_filter", False)):
        s = G.new_solver(0, network_type=ntype)
        rec = {"clean": clean, "label": label, "noisy": noisy, "img_cfg": G.CFG_CH_MSE, "seg_cfg": G.CFG_SP_CE,
               "network_type": ntype, "separate_training": sep}
        G.ref_step(s, clean, label, noisy, G.CFG_CH_MSE, G.CFG_SP_CE, rec, separate_training=sep)
        cases[name] = rec
        print("wrote", name, rec["losses"].tolist(), flush=True)

    # ---- R: un-injected scheme / k draws from seeded host RNGs
    c3, l3, n3 = synthetic_batch(3, 48, 48, seed=2)
    s = G.new_solver(0)
    random.seed(11)
    np.random.seed(11)
    s.train()
    s.reset_all_optimizers()
    s.standard_training(c3, l3, perturbed_image=n3)
    R = {"clean": c3, "label": l3, "noisy": n3, "img_cfg": CFG_RAND_MSE, "seg_cfg": CFG_RAND_CE, "seed": 11, "calls": []}
    zshape = tuple(s.z_i.shape)
    for _ in range(6):
        rec = {}
        masks, ks, keeps = [], [], []
        orig_perturb, orig_rand, orig_d2d = s.perturb_latent_code, np.random.rand, G.ref_model.F.dropout2d

        def perturb(*a, **kw):
            z, m = orig_perturb(*a, **kw)
            masks.append(m.detach().clone())
            return z, m

        def rand(*a):
            v = orig_rand(*a)
            ks.append(float(v))
            return v

        def d2d(inp, p=0.5, *a, **kw):
            out = orig_d2d(inp, p, *a, **kw)
            keeps.append(((out != 0).flatten(2).any(2) | (inp == 0).flatten(2).all(2)).float())
            return out

        s.perturb_latent_code, G.ref_mu.np.random.rand, G.ref_model.F.dropout2d = perturb, rand, d2d
        try:
            xh, yh = s.hard_example_generation(c3.clone(), l3.clone(), corrupted_image_DA_config=CFG_RAND_MSE,
                                               corrupted_seg_DA_config=CFG_RAND_CE)
        finally:
            s.perturb_latent_code, G.ref_mu.np.random.rand, G.ref_model.F.dropout2d = orig_perturb, orig_rand, orig_d2d
        rec["schemes"] = [scheme_of(m, zshape) for m in masks]
        rec["masks"] = [m if sc != "dropout" else None for m, sc in zip(masks, rec["schemes"])]
        rec["rand_draws"], rec["dropout_keeps"] = ks, keeps
        rec["x_hard"], rec["y_hard"] = xh.detach().clone(), yh.detach().clone()
        R["calls"].append(rec)
    R["buffers_after_calls"] = G.buffer_dump(s)
    # one full step whose two schemes are both targeted (hard masks: no torch draw at all), found by scanning seeds
    for seed in range(100):
        random.seed(seed)
        a = ["dropout", "spatial", "channel"]
        random.shuffle(a)
        b = ["dropout", "spatial", "channel"]
        random.shuffle(b)
        if a[0] != "dropout" and b[0] != "dropout" and a[0] != b[0]:
            break
    s = G.new_solver(0)
    random.seed(seed)
    np.random.seed(seed)
    step = {"seed": seed}
    G.ref_step(s, c3, l3, n3, CFG_RAND_MSE, CFG_RAND_CE, step)
    step["schemes"] = [scheme_of(m, zshape) for m in step["masks"]]
    R["step"] = step
    cases["R_random_scheme"] = R
    print("wrote R_random_scheme", [c["schemes"] for c in R["calls"]], step["schemes"], step["rand_draws"], flush=True)

    out = os.path.join(G.OUT, "cases_r2.pt")
    torch.save(cases, out)
    print("size:", os.path.getsize(out))


if __name__ == "__main__":
    main()
