#!/usr/bin/env python3
"""Adds the fp64 yardstick to the metric-sized records of tests/golden/cases_r2.pt:  python tools/gen_golden_r2_fp64.py

The reference's OWN fp32 gradients at bs16 x 256^2 are up to 7 % (checksums, relative to the tensor's norm) / 1 % (element-wise
relative L2) away from an fp64 evaluation of the same step (LeakyReLU-derivative ties and cancellation over 1 M pixels), so a
gradient comparison needs the fp64 value as the yardstick: error(HIP vs fp64) is judged against error(reference fp32 vs fp64) per
parameter.  The fp64 numbers come from oracle/ref_cpu.py in double precision (the oracle is pinned to the reference in fp32 by
tests/test_golden_r2.py); it takes ~2 min per record on 8 threads, hence a fixture.  Adds `grad_stats_64` and `grads_64`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_cpu as O      # noqa: E402
import test_golden_r2 as T           # noqa: E402

torch.set_num_threads(8)
path = os.path.join(ROOT, "tests", "golden", "cases_r2.pt")
r2 = torch.load(path, weights_only=False)
sd = torch.load(os.path.join(ROOT, "tests", "golden", "state_dicts_seed0.pt"), weights_only=False)
for case in ("H_bs16_dropout_step", "I_bs16_targeted_step"):
    rec = r2[case]
    o64 = O.OracleSolver(state_dicts=sd).double()
    clean, label, noisy = T.batch_of(rec)
    ov = T.overrides(rec, to=lambda t: t.double() if t.is_floating_point() else t)
    for o, m in zip(ov, rec["masks"]):
        if m is not None:
            o["mask"] = m          # same hard examples as the fp32 run (a near-tie in the fp64 ranking must not change the selection)
    losses = o64.cooperative_step(clean.double(), label, noisy.double(), rec["img_cfg"], rec["seg_cfg"], image_override=ov[0],
                                  seg_override=ov[1], do_optim=False)
    assert max(abs(a - float(b)) for a, b in zip(losses, rec["losses"])) < 1e-4, (losses, rec["losses"])
    rec["grad_stats_64"] = {f"{k}/{n}": (None if p.grad is None else T.stats(p.grad)) for k, m in o64.model.items() for n, p in m.named_parameters()}
    rec["grads_64"] = {key: dict(o64.model[key.split("/")[0]].named_parameters())[key.split("/")[1]].grad.clone() for key in rec["grads"]}
    print("added fp64 yardstick to", case, flush=True)
torch.save(r2, path)
print("size:", os.path.getsize(path))
