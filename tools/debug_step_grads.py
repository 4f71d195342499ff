"""GPU debug helper: for the golden full-step cases, relative L2 error of gradients
   HIP(fp32) vs oracle(fp64)   and   reference golden(fp32) vs oracle(fp64)
so that activation-tie noise (present in BOTH fp32 runs) can be told from a real bug."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import ref_cpu as O
from test_engine_gpu import _solver, _overrides, dev, is_dead_bias
cases = torch.load("tests/golden/cases.pt", weights_only=False)
sd = torch.load("tests/golden/state_dicts_seed0.pt", weights_only=False)
def to64(ov): return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ov.items()}
for case in ["C_step_channel_spatial", "D_step_dropout", "E_step_soft_random"]:
    C = cases[case]
    s = _solver(sd)
    ov_img, ov_seg = _overrides(C, (C["img_cfg"], C["seg_cfg"]))
    losses = s.cooperative_step(dev(C["clean"]), dev(C["label"]), dev(C["noisy"]), C["img_cfg"], C["seg_cfg"],
                                image_override=ov_img, seg_override=ov_seg, do_optim=False)
    o = O.OracleSolver(state_dicts=sd).double()
    oi, os_ = [to64({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in ov.items()}) for ov in (ov_img, ov_seg)]
    lo = o.cooperative_step(C["clean"].double(), C["label"], C["noisy"].double(), C["img_cfg"], C["seg_cfg"],
                            image_override=oi, seg_override=os_, do_optim=False)
    lh = torch.stack([v.detach().float() for v in losses]).cpu().double()
    print("====", case, "| loss err hip-vs-f64 %.2e  ref-vs-f64 %.2e" % (float((lh - torch.tensor(lo)).abs().max()),
          float((C["losses"] - torch.tensor(lo)).abs().max())),
          "| masks equal:", torch.equal(s.last_masks["image"].cpu().double(), o.last_masks["image"]) if C["img_cfg"]["mask_type"] != "dropout" else "n/a")
    for key, gref in C["grads"].items():
        k, n = key.split("/")
        if is_dead_bias(n): continue
        g64 = dict(o.model[k].named_parameters())[n].grad
        gh = dict(s.model[k].named_parameters())[n].grad.detach().cpu().double()
        print(f"  {key:58s} hip-vs-f64 {float((gh-g64).norm()/g64.norm()):.2e}   ref-vs-f64 {float((gref.double()-g64).norm()/g64.norm()):.2e}")
