"""The engine driven EXACTLY as the reference's caller drives its solver (VERDICT r2 item 2c / INTEGRATION.md "swap two imports, nothing
else"): the loop body of medseg/train_adv_supervised_segmentation_triplet.py:171-237 -- `train()`, `reset_all_optimizers()`,
`standard_training` -> `.item()` reads -> `reset_all_optimizers()` -> `hard_example_generation` -> `hard_example_training` (called
STAND-ALONE, `two_streams` at its default) -> `.item()` reads -> sum -> `reset_all_optimizers()` -> `loss.backward()` ->
`optimize_all_params()` -> `.item()`, with the `torch.cuda.empty_cache()` calls the script makes.  Checked against the reference's
recorded runs (goldens C, D at 2 x 64^2; H at bs16 x 256^2) and, bit for bit, against `cooperative_step` (the engine's own fused form
of the same sequence, which every other step test goes through)."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel  # noqa: E402
from test_engine_gpu import _overrides, _solver, close, dev  # noqa: E402
import test_golden_r2 as T2  # noqa: E402


def stock_loop_body(segmentation_solver, clean_image_l, label_l, image_l, corrupted_image_DA_config, corrupted_seg_DA_config,
                    latent_DA=True, separate_training=False, gen_corrupted_seg=True, gen_corrupted_image=True, inject=None):
    """train_adv_supervised_segmentation_triplet.py:171-237, statement for statement.  The input noise is supplied (the goldens recorded
    it, :185-187); `inject` carries the reference's recorded random draws into hard_example_generation (keyword-only extras of this
    build; the reference draws them from torch / numpy inside)."""
    device = clean_image_l.device
    loss_keys = ['loss/standard/total', 'loss/standard/seg', 'loss/standard/image', 'loss/standard/shape', 'loss/standard/gt_shape',
                 'loss/hard/total', 'loss/hard/seg', 'loss/hard/image', 'loss/hard/shape']
    loss_dict = {key: torch.tensor(0., device=device) for key in loss_keys}
    total_loss = 0.
    gc.collect()
    # step 1 (:176-177)
    segmentation_solver.train()
    segmentation_solver.reset_all_optimizers()
    # step 2: standard training (:191-199)
    seg_loss, image_recon_loss, gt_recon_loss, shape_recon_loss = segmentation_solver.standard_training(
        clean_image_l, label_l, perturbed_image=image_l, separate_training=separate_training)
    standard_loss = seg_loss + image_recon_loss + shape_recon_loss + gt_recon_loss
    loss_dict['loss/standard/total'] += standard_loss.item()
    loss_dict['loss/standard/seg'] += seg_loss.item()
    loss_dict['loss/standard/image'] += image_recon_loss.item()
    loss_dict['loss/standard/shape'] += shape_recon_loss.item()
    loss_dict['loss/standard/gt_shape'] += gt_recon_loss.item()
    hard = None
    if latent_DA:                                                                       # (:201-224)
        segmentation_solver.reset_all_optimizers()
        perturbed_image_0, perturbed_y_0 = segmentation_solver.hard_example_generation(
            clean_image_l.detach().clone(), label_l.detach().clone(), gen_corrupted_seg=gen_corrupted_seg,
            gen_corrupted_image=gen_corrupted_image, corrupted_image_DA_config=corrupted_image_DA_config,
            corrupted_seg_DA_config=corrupted_seg_DA_config, **(inject or {}))
        seg_supervised_loss, corrupted_image_recon_loss, shape_recon_loss_2, corrupted_shape_recon_loss = \
            segmentation_solver.hard_example_training(perturbed_image=perturbed_image_0, perturbed_seg=perturbed_y_0,
                                                      clean_image_l=clean_image_l, label_l=label_l, separate_training=separate_training)
        hard_loss = seg_supervised_loss + corrupted_image_recon_loss + shape_recon_loss_2 + corrupted_shape_recon_loss
        loss_dict['loss/hard/total'] += hard_loss.item()
        loss_dict['loss/hard/seg'] += seg_supervised_loss.item()
        loss_dict['loss/hard/image'] += corrupted_image_recon_loss.item()
        loss_dict['loss/hard/shape'] += (shape_recon_loss_2 + corrupted_shape_recon_loss).item()
        torch.cuda.empty_cache()
        hard = (seg_supervised_loss, corrupted_image_recon_loss, shape_recon_loss_2, corrupted_shape_recon_loss)
    else:
        hard_loss = torch.tensor(0., device=device)
    loss = standard_loss + hard_loss                                                    # (:228-233)
    segmentation_solver.reset_all_optimizers()
    loss.backward()
    segmentation_solver.optimize_all_params()
    total_loss += loss.item()
    torch.cuda.empty_cache()
    eight = [seg_loss, image_recon_loss, gt_recon_loss, shape_recon_loss] + (list(hard) if hard is not None else [hard_loss] * 4)
    return torch.stack([v.detach().float() for v in eight]).cpu().double(), loss_dict, total_loss


def _state(s):
    torch.cuda.synchronize()
    return ({k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()},
            {k: (m._bflat.detach().cpu().clone(), m._nbt.detach().cpu().clone()) for k, m in s.model.items()},
            {k: (o.exp_avg.cpu().clone(), o.exp_avg_sq.cpu().clone(), o.step_count) for k, o in s.optimizers.items()})


def _assert_same_state(a, b):
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), f"weights of {k}"
        assert torch.equal(a[1][k][0], b[1][k][0]) and torch.equal(a[1][k][1], b[1][k][1]), f"BatchNorm buffers of {k}"
        assert torch.equal(a[2][k][0], b[2][k][0]) and torch.equal(a[2][k][1], b[2][k][1]) and a[2][k][2] == b[2][k][2], f"Adam state of {k}"


@pytest.mark.parametrize("case", ["C_step_channel_spatial", "D_step_dropout"])
def test_stock_loop_vs_golden_and_cooperative_step(golden_cases, golden_sd, case):
    C = golden_cases[case]
    ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    clean, label, noisy = dev(C["clean"]), dev(C["label"]), dev(C["noisy"])
    s = _solver(golden_sd)
    assert s.two_streams                                                        # the default: hard_example_training forks its image branch itself
    got, loss_dict, total = stock_loop_body(s, clean, label, noisy, C["img_cfg"], C["seg_cfg"], inject={"image_override": ov_img, "seg_override": ov_seg})
    assert torch.allclose(got, C["losses"], atol=1e-4, rtol=0), (got, C["losses"])
    assert abs(total - float(C["losses"].sum())) < 4e-4 and abs(float(loss_dict['loss/hard/total']) - float(C["losses"][4:].sum())) < 2e-4
    if C["img_cfg"]["mask_type"] != "dropout":
        assert torch.equal(s.last_masks["image"].cpu(), C["masks"][0]) and torch.equal(s.last_masks["seg"].cpu(), C["masks"][1])
    for key, b in C["buffers_after"].items():
        k, n = key.split("/")
        close(dict(s.model[k].named_buffers())[n].double(), b.double(), atol=2e-5, rel=1e-5, what=key)
    # the engine's fused form of the same sequence: bit for bit the same training state, over TWO consecutive iterations
    # (with one backward per pass, as the stock loop's autograd sweep runs it; the fused step's default -- the standard and the hard pass
    #  of a network stacked in one backward, solver.stack_passes -- changes summation orders only: tests/test_stack_gpu.py)
    ref = _solver(golden_sd)
    ref.stack_passes = ()
    l_ref = ref.cooperative_step(clean, label, noisy, C["img_cfg"], C["seg_cfg"], image_override=ov_img, seg_override=ov_seg)
    assert torch.equal(torch.stack([v.detach().float() for v in l_ref]).cpu().double(), got)
    _assert_same_state(_state(ref), _state(s))
    got2, _, _ = stock_loop_body(s, clean, label, noisy, C["img_cfg"], C["seg_cfg"], inject={"image_override": ov_img, "seg_override": ov_seg})
    l_ref2 = ref.cooperative_step(clean, label, noisy, C["img_cfg"], C["seg_cfg"], image_override=ov_img, seg_override=ov_seg)
    assert torch.equal(torch.stack([v.detach().float() for v in l_ref2]).cpu().double(), got2)
    _assert_same_state(_state(ref), _state(s))


def test_stock_loop_at_bs16_256_vs_reference(golden_sd):
    """Golden H (BASELINE configs[1]: bs16 x 256^2, dropout masks) through the stock loop: the reference's 8 losses, BatchNorm buffers,
    every parameter gradient by random projections and the direction of every Adam update; bitwise `cooperative_step`."""
    import os
    r2 = torch.load(os.path.join(T2.HERE, "golden", "cases_r2.pt"), weights_only=False)
    r3 = torch.load(os.path.join(T2.HERE, "golden", "cases_r3.pt"), weights_only=False)
    rec, rec3 = r2["H_bs16_dropout_step"], r3["H_bs16_dropout_step"]
    clean, label, noisy = (dev(t) for t in T2.batch_of(rec))
    ov = T2.overrides(rec, to=lambda t: t.to("cuda"))
    s = _solver(golden_sd)
    grads = {}
    orig = s.optimize_all_params

    def optimize_all_params():                               # (between loss.backward() and the Adam step: this iteration's gradients)
        grads.update({f"{k}/{n}": p.grad.detach().clone() for k, m in s.model.items() for n, p in m.named_parameters()})
        orig()
    s.optimize_all_params = optimize_all_params
    got, _, _ = stock_loop_body(s, clean, label, noisy, rec["img_cfg"], rec["seg_cfg"], inject={"image_override": ov[0], "seg_override": ov[1]})
    s.optimize_all_params = orig
    T2._check_hip_step(rec, s, got, grads, grad_rtol=1e-2, yardstick=rec["grad_stats_64"], rec3=rec3, sd_before=golden_sd)
    ref = _solver(golden_sd)
    ref.stack_passes = ()
    l_ref = ref.cooperative_step(clean, label, noisy, rec["img_cfg"], rec["seg_cfg"], image_override=ov[0], seg_override=ov[1])
    assert torch.equal(torch.stack([v.detach().float() for v in l_ref]).cpu().double(), got)
    _assert_same_state(_state(ref), _state(s))


def test_hard_example_training_stand_alone_matches_oracle(golden_cases, golden_sd):
    """`hard_example_training` called directly (model.py:525-559) on given hard examples: 4 losses and, after backward, the input-side
    gradient norms against the CPU oracle -- with two_streams on (the default: the image branch forks inside the call) and off."""
    C = golden_cases["C_step_channel_spatial"]
    clean, label, noisy = C["clean"], C["label"], C["noisy"]
    xh = torch.rand(clean.shape, generator=torch.Generator().manual_seed(4))
    yh = torch.randn(clean.shape[0], 4, *clean.shape[2:], generator=torch.Generator().manual_seed(5))
    o = O.OracleSolver(state_dicts=golden_sd)
    o.train()
    o.reset_all_optimizers()
    lo = o.hard_example_training(xh, clean, yh, label)
    sum(lo).backward()
    outs = []
    for two in (True, False):
        s = _solver(golden_sd)
        s.two_streams = two
        s.train()
        s.reset_all_optimizers()
        lh = s.hard_example_training(perturbed_image=dev(xh), clean_image_l=dev(clean), perturbed_seg=dev(yh), label_l=dev(label))
        (lh[0] + lh[1] + lh[2] + lh[3]).backward()
        torch.cuda.synchronize()
        got = torch.stack([v.detach().float() for v in lh]).cpu()
        assert torch.allclose(got, torch.stack([v.detach() for v in lo]), atol=1e-4, rtol=0), (got, lo)
        for k, m in s.model.items():
            for n, p in m.named_parameters():
                po = dict(o.model[k].named_parameters())[n]
                if T2.is_dead_bias(n) or po.grad is None:
                    continue
                nh, no = float(p.grad.norm()), float(po.grad.norm())
                assert abs(nh - no) <= 5e-2 * no + 1e-7, (k, n, nh, no)
        outs.append((got, {k: m._flat.grad.detach().cpu().clone() for k, m in s.model.items()}))
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(outs[0][1][k], outs[1][1][k]) for k in outs[0][1])
