"""How close are the top-k selections of the targeted latent masks to a tie in the recorded full-size step of the reference
(tests/golden/cases_r2.pt, I_bs16_targeted_step)?  CPU only: the oracle's step with rank_select_mask wrapped; prints, per mask call, the
smallest relative distance between the threshold score and its neighbours on either side (an element closer than fp32 summation noise,
~1e-7, can land on either side of the threshold in ANY fp32 implementation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import ref_cpu as O
import test_golden_r2 as T

r2 = torch.load(os.path.join(os.path.dirname(T.__file__), "golden", "cases_r2.pt"), weights_only=False)
sd = torch.load(os.path.join(os.path.dirname(T.__file__), "golden", "state_dicts_seed0.pt"), weights_only=False)
orig = O.rank_select_mask
calls = []


def wrapped(score, k, soft_noise):
    srt = torch.sort(score, dim=1, descending=True)[0]
    thr = srt[:, k]
    above = srt[:, k - 1] if k > 0 else None
    below = srt[:, k + 1] if k + 1 < srt.shape[1] else None
    scale = score.abs().max(dim=1)[0]
    ga = ((above - thr) / scale) if above is not None else torch.full_like(thr, float("inf"))
    gb = ((thr - below) / scale) if below is not None else torch.full_like(thr, float("inf"))
    calls.append((tuple(score.shape), k, float(ga.min()), float(gb.min()), int(torch.argmin(torch.minimum(ga, gb)))))
    return orig(score, k, soft_noise)


O.rank_select_mask = wrapped
case = sys.argv[1] if len(sys.argv) > 1 else "I_bs16_targeted_step"
s, losses = T._oracle_step(r2[case], sd)
for shape, k, ga, gb, img in calls:
    print(f"{case}: scores {shape}, k = {k}: smallest gap (threshold element to its neighbours, relative to max|score| of the image): "
          f"above {ga:.2e}, below {gb:.2e} (image {img})")
