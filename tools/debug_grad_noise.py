"""Per-parameter gradient error of the HIP step and of the reference's own fp32 run against the fp64 yardstick (bs16 x 256^2 goldens)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_golden_r2 as T
r2 = torch.load(os.path.join(ROOT, "tests", "golden", "cases_r2.pt"), weights_only=False)
sd = torch.load(os.path.join(ROOT, "tests", "golden", "state_dicts_seed0.pt"), weights_only=False)
for case in sys.argv[1:] or ["H_bs16_dropout_step", "I_bs16_targeted_step"]:
    rec = r2[case]
    s, got, grads = T._hip_step(rec, sd)
    rows = []
    for key, e in rec["grad_stats"].items():
        if e is None or T.is_dead_bias(key): continue
        y = rec["grad_stats_64"][key]
        sh = T.stats(grads[key])
        scale = torch.stack([torch.maximum(y[0].abs(), y[1]), y[1], y[2]]).clamp_min(1e-12)
        rows.append((key, ((sh - y).abs() / scale).tolist(), ((e - y).abs() / scale).tolist()))
    for i, nm in enumerate(("sum", "l2", "absmax")):
        hip = sorted(((r[1][i], r[2][i], r[0]) for r in rows), reverse=True)
        print(case, nm, "max hip %.2e  max ref %.2e  median hip %.2e median ref %.2e" % (hip[0][0], max(r[2][i] for r in rows), hip[len(hip)//2][0], sorted(r[2][i] for r in rows)[len(rows)//2]))
        for h in hip[:5]: print("    hip %.2e ref %.2e %s" % h)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for key, gref in rec["grads"].items():
        if gref is None or T.is_dead_bias(key): continue
        g64 = rec["grads_64"][key]
        print("   picked %-55s hip %.2e ref %.2e" % (key, rel(grads[key].cpu().double(), g64), rel(gref.double(), g64)))
