"""Latent-mask kernels (score + rank-select + apply) against the HBM roofline: the configured size (16x128x16x16, which
lives in L2 / Infinity Cache and is launch-latency bound) and a sweep into the HBM-bound regime.
Algorithmic bytes = 3 * N*C*H*W*4 (read grad, read code, write masked) + score/mask vectors.   python tools/bench_mask.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import ops

PEAK = 8000.0
def run(n, c, h, w, mode, iters=20):
    grad = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    code = torch.rand(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    L = c if mode == 0 else h * w
    k = L // 3
    for _ in range(3):
        s = ops.latent_score(grad, mode); ops.latent_mask_apply(code, s, mode, k)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(iters): s = ops.latent_score(grad, mode)
    e[1].record()
    for _ in range(iters): ops.latent_mask_apply(code, s, mode, k)
    e[2].record(); torch.cuda.synchronize()
    t_score, t_apply = e[0].elapsed_time(e[1]) * 1e3 / iters, e[1].elapsed_time(e[2]) * 1e3 / iters
    for _ in range(3): ops.latent_mask(grad, code, mode, k)
    torch.cuda.synchronize()
    f = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    f[0].record()
    for _ in range(iters): ops.latent_mask(grad, code, mode, k)
    f[1].record(); torch.cuda.synchronize()
    t_fused = f[0].elapsed_time(f[1]) * 1e3 / iters
    elems = n * c * h * w
    b_score, b_apply = 4 * elems + 4 * n * L, 8 * elems + 8 * n * L
    extra = {}
    if elems * 4 <= (8 << 20):          # launch-latency-bound sizes: the GPU side alone, replayed from a HIP graph
        def graph_us(fn):
            g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(10): fn()
            torch.cuda.current_stream().wait_stream(side)
            g.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): g.replay()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / 100
        extra = {"graph_fused_us": round(graph_us(lambda: ops.latent_mask(grad, code, mode, k)), 2),
                 "graph_3launch_us": round(graph_us(lambda: ops.latent_mask_apply(code, ops.latent_score(grad, mode), mode, k)), 2)}
        extra["graph_fused_frac_of_8TBs"] = round((b_score + b_apply) / extra["graph_fused_us"] / 1e3 / PEAK, 3)
    return {**extra, "fused_us": round(t_fused, 1), "fused_GBs": round((b_score + b_apply) / t_fused / 1e3, 1), "fused_frac_of_8TBs": round((b_score + b_apply) / t_fused / 1e3 / PEAK, 3),
            "shape": [n, c, h, w], "mode": "channel" if mode == 0 else "spatial", "tensor_MiB": round(elems * 4 / 2**20, 1),
            "score_us": round(t_score, 1), "apply_us": round(t_apply, 1), "total_us": round(t_score + t_apply, 1),
            "score_GBs": round(b_score / t_score / 1e3, 1), "apply_GBs": round(b_apply / t_apply / 1e3, 1),
            "total_GBs": round((b_score + b_apply) / (t_score + t_apply) / 1e3, 1),
            "frac_of_8TBs": round((b_score + b_apply) / (t_score + t_apply) / 1e3 / PEAK, 3)}

if __name__ == "__main__":
    shapes = [(16, 128, 16, 16), (16, 128, 64, 64), (64, 128, 64, 64), (64, 128, 128, 128), (128, 128, 128, 128)]
    out = []
    for shp in shapes:
        for mode in (0, 1):
            if mode == 1 and shp[2] * shp[3] > 8192: continue        # spatial rows are capped at 8192 positions
            r = run(*shp, mode); out.append(r); print(json.dumps(r))
    os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/latent_mask_sweep.json", "w"), indent=1)
