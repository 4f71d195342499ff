"""Worker of tests/test_dist_gpu.py: one data-parallel rank running REAL HIP cooperative steps on its shard (all ranks on GPU 0 over
gloo when the box has one GPU; RCCL for the world-1 smoke).  Writes what the parent compares to <out>/rank<r>.pt."""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as O                                                                  # noqa: E402  (synthetic inputs only)
from cooperative_training_and_latent_space_data_augmentation_amd.dist import DataParallel        # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel      # noqa: E402

CH_MSE = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
SP_CE = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
DROP_MSE = {"loss_name": "mse", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
DROP_CE = {"loss_name": "ce", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}


def shard(rank, device):
    c, l, n = O.synthetic_batch(2, 64, 64, seed=50 + rank)
    dev = lambda t: (t.to(device).contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.to(device))
    return dev(c), dev(l), dev(n)


def _weights(s):
    torch.cuda.synchronize()
    return {k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()}


def graph_main(rank, world, device, out, sd):
    """graph-mode data parallelism (the five all-reduces run eagerly between the forward/backward graph and the Adam graph)
    must end at bitwise the eager-DP weights; with mask_type='random' (BASELINE configs[3]: per-rank scheme draws, one graph per scheme pair)
    every rank must still hold the same weights after every step."""
    from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
    CH_RT = dict(CH_MSE, random_threshold=True)
    SP_RT = dict(SP_CE, random_threshold=True)
    RND_MSE = {"loss_name": "mse", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
    RND_CE = {"loss_name": "ce", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
    clean, label, noisy = shard(rank, device)
    rec = {}
    for tag, use_graph in (("eager", False), ("graph", True)):
        torch.manual_seed(100 + rank)
        s = AdvancedTripletReconSegmentationModel(use_gpu=True)
        if rank == 0:
            for k, m in s.model.items():
                m.load_state_dict(sd[k])
        dp = DataParallel(s)
        torch.manual_seed(7000 + rank); np.random.seed(7000 + rank); random.seed(7000 + rank)
        g = CooperativeStepGraph(s, CH_RT, SP_RT, grad_hook=dp.sync_gradients) if use_graph else None
        losses = []
        for _ in range(3):
            l = g(clean, label, noisy) if use_graph else s.cooperative_step(clean, label, noisy, CH_RT, SP_RT, grad_hook=dp.launch_remaining)
            losses.append(torch.stack([v.detach().float() for v in l]).cpu())
        rec[tag] = {"weights": _weights(s), "losses": losses, "k_next": float(np.random.rand())}
    # configs[3]: all three schemes randomly sampled per rank and step, graph mode
    torch.manual_seed(100 + rank)
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    if rank == 0:
        for k, m in s.model.items():
            m.load_state_dict(sd[k])
    dp = DataParallel(s)
    torch.manual_seed(7000 + rank); np.random.seed(7000 + rank); random.seed(7000 + rank)
    g = CooperativeStepGraph(s, RND_MSE, RND_CE, grad_hook=dp.sync_gradients, replay="segments")      # (the segment replay under DP as well)
    per_step = []
    for _ in range(4):
        l = g(clean, label, noisy)
        per_step.append((_weights(s), torch.stack([v.detach().float() for v in l]).cpu()))
    rec["random"] = {"per_step": per_step, "schemes": sorted(g.entries.keys()), "replays": g.replays}
    torch.save(rec, os.path.join(out, f"graph_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def world4_main(rank, world, device, out, sd):
    """more than two ranks.  With > 2 ranks a ring all-reduce ties an element's summation order to its position in
    the buffer, so the five per-network ranges need not sum to the single bucket's bits -- what must hold is that every rank ends with the
    SAME bits.  Three steps (per-network exchange launched from inside backward, then a dropout step, then targeted with random k),
    per-rank shards and RNG streams; every rank saves its weights after each step, and the gradient bucket of the first."""
    CH_RT = dict(CH_MSE, random_threshold=True)
    SP_RT = dict(SP_CE, random_threshold=True)
    torch.manual_seed(100 + rank)
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    if rank == 0:
        for k, m in s.model.items():
            m.load_state_dict(sd[k])
    dp = DataParallel(s)
    torch.manual_seed(7000 + rank); np.random.seed(7000 + rank); random.seed(7000 + rank)
    clean, label, noisy = shard(rank, device)
    rec = {"per_step": [], "world": world}

    def hook(solver):
        dp.sync_gradients(solver)
        rec["bucket_sum"] = dp.bucket.buf.detach().cpu().clone()
        rec["launched_in_backward"] = list(dp.launched_in_backward)

    for i, (ci, cs, h) in enumerate(((CH_MSE, SP_CE, hook), (DROP_MSE, DROP_CE, dp.launch_remaining), (CH_RT, SP_RT, dp.launch_remaining))):
        l = s.cooperative_step(clean, label, noisy, ci, cs, grad_hook=h)
        rec["per_step"].append((_weights(s), torch.stack([v.detach().float() for v in l]).cpu()))
    # this rank's own (un-reduced) gradient of step 1, for the parent's sum check: one more solver from the broadcast weights
    torch.save(rec, os.path.join(out, f"w4_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    backend, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    sd = torch.load(os.path.join(ROOT, "tests", "golden", "state_dicts_seed0.pt"), weights_only=False)
    if len(sys.argv) > 3 and sys.argv[3] == "graph":
        return graph_main(rank, world, device, out, sd)
    if len(sys.argv) > 3 and sys.argv[3] == "world4":
        return world4_main(rank, world, device, out, sd)
    torch.manual_seed(100 + rank)                      # deliberately different weights before the broadcast
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    if rank == 0:
        for k, m in s.model.items():
            m.load_state_dict(sd[k])
    dp = DataParallel(s)
    # per-rank RNG streams: seeded after the weight broadcast
    torch.manual_seed(7000 + rank); np.random.seed(7000 + rank); random.seed(7000 + rank)
    clean, label, noisy = shard(rank, device)
    rec = {}

    def hook(solver):
        dp.sync_gradients(solver)                # launches what backward has not launched yet, then waits for all five ranges
        rec["bucket_sum"] = dp.bucket.buf.detach().cpu().clone()
        rec["launched_in_backward"] = list(dp.launched_in_backward)

    losses = s.cooperative_step(clean, label, noisy, CH_MSE, SP_CE, grad_hook=hook)
    torch.cuda.synchronize()
    rec["losses"] = torch.stack([v.detach().float() for v in losses]).cpu()
    rec["weights"] = {k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()}
    rec["buffers"] = {k: (m._bflat.detach().cpu().clone(), m._nbt.detach().cpu().clone()) for k, m in s.model.items()}
    s.cooperative_step(clean, label, noisy, DROP_MSE, DROP_CE, grad_hook=dp.launch_remaining)   # per-rank dropout patterns; every Adam launch
    rec["launched_in_backward2"] = list(dp.launched_in_backward)                                # waits for its own range (optimize_all_params)
    rec["drop_masks"] = {k: v.detach().cpu().clone() for k, v in s.last_masks.items()}
    rec["weights2"] = {k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()}
    torch.cuda.synchronize()
    torch.save(rec, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
