"""Round 6: hunt for the NaN seen (about 1 run in 10, round-5 code included) in `bench.py --gpus 2 --backend gloo --all-on-device0`.
Two (or more) processes share GPU 0; each runs cooperative steps at 64x64 and checks losses / weights / gradients for non-finite values
after EVERY step, naming the first offender.  Variants separate the suspects: data parallelism on / off, exchange launched from inside
backward or behind it, two launch chains or one, eager / graph / segments.

    python tools/debug/r6_dp_stress.py [variant ...]         (parent: spawns the ranks per variant)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANTS = {  # name: (dp, overlap, two_streams, mode, steps)
    "dp_eager": (True, True, True, "eager", 150),
    "nodp_eager": (False, False, True, "eager", 150),
    "dp_eager_no_overlap": (True, False, True, "eager", 150),
    "dp_eager_one_chain": (True, True, False, "eager", 150),
    "dp_graph": (True, True, True, "graph", 100),
    "dp_segments": (True, True, True, "segments", 100),
    "dp_mixed": (True, True, True, "mixed", 120),
}


def child(name):
    import torch
    import torch.distributed as dist
    from oracle import ref_cpu as O
    from cooperative_training_and_latent_space_data_augmentation_amd.dist import DataParallel
    from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
    from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
    use_dp, overlap, two, mode, steps = VARIANTS[name]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    s.two_streams = two
    dp = DataParallel(s, overlap=overlap) if use_dp else None
    c, l, n = O.synthetic_batch(2, 64, 64, seed=50 + rank)
    dev = lambda t: (t.cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.cuda())
    c, l, n = dev(c), dev(l), dev(n)
    drop_i = {"loss_name": "mse", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
    drop_s = {"loss_name": "ce", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
    hook = (dp.launch_remaining if overlap else dp.sync_gradients) if dp else None
    g = None
    if mode in ("graph", "segments", "mixed"):
        g = CooperativeStepGraph(s, drop_i, drop_s, grad_hook=dp.sync_gradients if dp else None, replay="segments" if mode == "segments" else "runtime")

    def check(i, what, losses):
        torch.cuda.synchronize()
        bad = []
        if not all(bool(torch.isfinite(v)) for v in losses):
            bad.append("losses " + str([round(float(v), 4) for v in losses]))
        for k, m in s.model.items():
            if not bool(torch.isfinite(m._flat_data).all()):
                bad.append(f"weights of {k}")
            if not bool(torch.isfinite(m._flat.grad).all()):
                bad.append(f"gradient of {k}")
            if not bool(torch.isfinite(m._bflat).all()):
                bad.append(f"BatchNorm buffers of {k}")
        if bad:
            print(f"[{name} rank {rank}] step {i} ({what}): NON-FINITE: {bad}", flush=True)
            return False
        return True
    ok = True
    t0 = time.time()
    for i in range(steps):
        if mode == "eager" or (mode == "mixed" and (i // 8) % 2 == 0):
            what, losses = "eager", s.cooperative_step(c, l, n, drop_i, drop_s, grad_hook=hook)
        else:
            if mode == "mixed":
                g.set_replay_mode("segments" if (i // 16) % 2 else "runtime")
            what, losses = g.replay_mode, g(c, l, n)
        if not check(i, what, losses):
            ok = False
            break
    print(f"[{name} rank {rank}] {'ok' if ok else 'FAILED'}: {i + 1} steps in {time.time() - t0:.1f} s", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        names = sys.argv[1:] or list(VARIANTS)
        world = int(os.environ.get("STRESS_WORLD", "2"))
        for k, name in enumerate(names):
            procs = []
            for r in range(world):
                env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + k), HSA_ENABLE_IPC_MODE_LEGACY="0")
                procs.append(subprocess.Popen([sys.executable, __file__, "child", name], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            for p in procs:
                out, _ = p.communicate(timeout=1500)
                for line in out.splitlines():
                    if line.startswith("[") or "Error" in line or "error" in line:
                        print(line[:400], flush=True)
