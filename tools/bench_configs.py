"""Secondary measurements for DESIGN.md (1 GPU): BASELINE.json configs other than the headline one.
   python tools/bench_configs.py            -> JSON lines (slices/s)
config 3a/3b: bs16 256x256 full cooperative step with targeted masks (channel+mse on z_i, spatial+ce on z_s and swapped), fp32
config 4'   : bs16, mask_type='random' (python RNG picks dropout/spatial/channel per code), single rank
config 5    : inference, 10-slice chunks of 192x192, predict(n_iter=1|2), eval-mode BatchNorm, + argmax
   python tools/bench_configs.py --cpu      additionally times the CPU oracle (oracle/ref_cpu.py, 32 threads) on configs 3 and 5
                                            (config 3: 1 warm-up + 3 steps, median; config 5: 1 warm-up + 5 chunks, median)"""
import json, os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd import ops

dev = torch.device("cuda")
def cfg(loss, kind, rnd=True, soft=True):
    return {"loss_name": loss, "mask_type": kind, "max_threshold": 0.5, "random_threshold": rnd, "if_soft": soft}
def synth(n, h, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 1, h, h, generator=g); l = torch.randint(0, 4, (n, h, h), generator=g)
    return c.to(dev), l.to(dev), torch.clamp(c + 0.05 * torch.randn(n, 1, h, h, generator=g), 0, 1).to(dev)

torch.manual_seed(0); np.random.seed(0); random.seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)
clean, label, noisy = synth(16, 256, 1)
def time_steps(img_cfg, seg_cfg, steps=10, warm=3):
    for _ in range(warm): s.cooperative_step(clean, label, noisy, img_cfg, seg_cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): s.cooperative_step(clean, label, noisy, img_cfg, seg_cfg)
    torch.cuda.synchronize(); return 16 * steps / (time.perf_counter() - t0)
def time_graph(img_cfg, seg_cfg, solver, steps=10, warm=3):
    from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
    for _ in range(warm): solver.cooperative_step(clean, label, noisy, img_cfg, seg_cfg)
    g = CooperativeStepGraph(solver, img_cfg, seg_cfg)
    for _ in range(warm): g(clean, label, noisy)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): g(clean, label, noisy)
    torch.cuda.synchronize(); return 16 * steps / (time.perf_counter() - t0)
for name, ic, sc in [("config2_dropout", cfg("mse", "dropout"), cfg("ce", "dropout")),
                     ("config3a_channel_mse+spatial_ce", cfg("mse", "channel"), cfg("ce", "spatial")),
                     ("config3b_spatial_mse+channel_ce", cfg("mse", "spatial"), cfg("ce", "channel")),
                     ("config4_random_scheme_single_rank", cfg("mse", "random"), cfg("ce", "random"))]:
    rec = {"config": name, "slices_per_s": round(time_steps(ic, sc), 1), "dtype": "f32", "batch": 16, "size": 256}
    if "random" not in name:          # (the random scheme picks a scheme per step on the host: one graph per scheme, not timed here)
        rec["slices_per_s_graph_replay"] = round(time_graph(ic, sc, s), 1)
    print(json.dumps(rec), flush=True)
# config 3 as BASELINE states it: bf16 storage + bf16 MFMA
s16 = AdvancedTripletReconSegmentationModel(use_gpu=True, compute_dtype="bf16")
_s, s = s, s16
for name, ic, sc in [("config3a_channel_mse+spatial_ce", cfg("mse", "channel"), cfg("ce", "spatial")),
                     ("config3b_spatial_mse+channel_ce", cfg("mse", "spatial"), cfg("ce", "channel"))]:
    print(json.dumps({"config": name, "slices_per_s": round(time_steps(ic, sc), 1), "slices_per_s_graph_replay": round(time_graph(ic, sc, s16), 1),
                      "dtype": "bf16", "batch": 16, "size": 256}), flush=True)
s = _s
vol = torch.rand(10, 1, 192, 192, device=dev)
for n_iter in (1, 2):
    for _ in range(3): ops.argmax_c(s.predict(vol, n_iter=n_iter))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): lab = ops.argmax_c(s.predict(vol, n_iter=n_iter))
    torch.cuda.synchronize()
    print(json.dumps({"config": f"config5_inference_192_n_iter{n_iter}", "slices_per_s": round(10 * 20 / (time.perf_counter() - t0), 1),
                      "chunk": 10, "size": 192}), flush=True)

if "--cpu" in sys.argv:
    from oracle import ref_cpu as O
    from cooperative_training_and_latent_space_data_augmentation_amd.init import reference_init_state_dicts
    threads = min(32, os.cpu_count())
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    o = O.OracleSolver(state_dicts=reference_init_state_dicts())
    hc, hl, hn = clean.cpu(), label.cpu(), noisy.cpu()
    ts = []
    for i in range(4):
        t0 = time.perf_counter(); o.cooperative_step(hc, hl, hn, cfg("mse", "channel"), cfg("ce", "spatial")); ts.append(time.perf_counter() - t0)
    med = sorted(ts[1:])[1]
    print(json.dumps({"config": "config3a_cpu_oracle", "slices_per_s": round(16 / med, 2), "cores": threads, "kind": "port",
                      "sample": f"1 warm-up + 3 steps, median {med:.1f} s (all {[round(t, 1) for t in ts[1:]]})"}), flush=True)
    o.eval()
    hv = vol.cpu()
    for n_iter in (1, 2):
        ts = []
        with torch.no_grad():
            for i in range(6):
                t0 = time.perf_counter()
                lab = o.predict(hv, n_iter=n_iter).argmax(1)
                ts.append(time.perf_counter() - t0)
        med = sorted(ts[1:])[2]
        print(json.dumps({"config": f"config5_cpu_oracle_n_iter{n_iter}", "slices_per_s": round(10 / med, 1), "cores": threads, "kind": "port",
                          "sample": f"10x192x192 chunk, 1 warm-up + 5 chunks, median {med * 1e3:.0f} ms"}), flush=True)
