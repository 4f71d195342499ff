"""Calibration of tests/test_bf16_backward_gpu.py: per network and BatchNorm mode, the relative L2 distance of every gradient of the HIP
bf16 engine from the plan emulation (oracle/bf16_plan.py), and, as the noise yardstick, of the fp64-arithmetic emulation from the
fp32-arithmetic one (same rounding points, different summation: what ANY two implementations differ by)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import bf16_plan as P, ref_cpu as O
from cooperative_training_and_latent_space_data_augmentation_amd import nets
from cooperative_training_and_latent_space_data_augmentation_amd.model_util import _disable_tracking_bn_stats
torch.set_num_threads(16)
sd = torch.load("tests/golden/state_dicts_seed0.pt", weights_only=False)
NET_INPUT = {"image_encoder": (1, 128, 128), "shape_encoder": (4, 128, 128), "segmentation_decoder": (128, 8, 8), "shape_decoder": (128, 8, 8), "image_decoder": (128, 8, 8)}
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
dead = ("conv.0.bias", "conv.3.bias", "inc.0.bias", "inc.3.bias", "final_conv.0.bias", "code_decoupler.0.bias", "code_decoupler.3.bias")
def nhwc(x): return x.cuda().contiguous(memory_format=torch.channels_last)
for name, (c, h, w) in NET_INPUT.items():
    for mode in ("A", "B"):
        g = torch.Generator().manual_seed(3)
        n = 4
        x = torch.relu(torch.randn(n, c, h, w, generator=g)) if "decoder" in name else torch.rand(n, c, h, w, generator=g)
        res = {}
        for dt in (torch.float32, torch.float64):
            onet = O.build_networks(init=False)[name]; onet.load_state_dict(sd[name]); onet = onet.to(dt)
            ctx = O.bn_no_track(onet) if mode == "B" else None
            if ctx: ctx.__enter__()
            outs, rec = P.net_forward(onet, x.to(dt))
            douts = [torch.randn(o.shape, generator=torch.Generator().manual_seed(5 + i)) for i, o in enumerate(outs)]
            dx, grads = P.net_backward(onet, rec, [d.to(dt) for d in douts])
            if ctx: ctx.__exit__(None, None, None)
            res[dt] = (outs, dx, grads)
        hnet = nets.build_networks(device="cuda", state_dicts={name: sd[name]}, dtype="bf16")[name]
        xh = nhwc(x).requires_grad_(True)
        if mode == "B":
            with _disable_tracking_bn_stats(hnet):
                yh = hnet(xh)
        else:
            yh = hnet(xh)
        yh = yh if isinstance(yh, tuple) else (yh,)
        torch.autograd.backward(yh, [nhwc(d) for d in douts])
        o32, dx32, g32 = res[torch.float32]
        o64, dx64, g64 = res[torch.float64]
        print(f"== {name} mode {mode}: out HIP-vs-emul {[round(rel(a.cpu(), b), 5) for a, b in zip(yh, o32)]} emul64-vs-emul32 {[round(rel(a, b), 5) for a, b in zip(o64, o32)]}")
        print(f"   dx  HIP {rel(xh.grad.cpu(), dx32):.4f}   emul64 {rel(dx64, dx32):.4f}")
        hp = dict(hnet.named_parameters())
        rows = []
        for k, gr in g32.items():
            if k.endswith(dead): continue
            rows.append((k, rel(hp[k].grad.cpu(), gr), rel(g64[k], gr)))
        worst = sorted(rows, key=lambda r: -r[1])[:6]
        print("   worst params (HIP, emul64):", [(k, round(a, 4), round(b, 4)) for k, a, b in worst])
        import statistics
        print("   median HIP", round(statistics.median(r[1] for r in rows), 5), "median emul64", round(statistics.median(r[2] for r in rows), 5), "n", len(rows))
        missing = [k for k, p in hp.items() if k not in g32 and p.grad is not None and float(p.grad.abs().max()) > 0 and not (mode == "B")]
        print("   params without emulated gradient:", missing[:5])
