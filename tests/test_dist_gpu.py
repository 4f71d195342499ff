"""Data parallelism on the real HIP step (SURVEY 8(e) "DP oracle"): N single-rank runs on the N shards with identical weights give the
expected all-reduced gradient; BatchNorm statistics stay rank-local; the update uses the mean gradient.  One GPU is enough: both ranks
run on GPU 0 over gloo (every kernel is deterministic, so the comparison is bit for bit).  Plus a world-1 RCCL smoke."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _launch(world, backend, out, port, *extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), backend, str(out), *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]


def test_two_rank_step_equals_mean_of_single_rank_gradients(golden_sd, tmp_path):
    import dp_worker as W
    from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
    _launch(2, "gloo", tmp_path, 29561)
    ranks = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(2)]
    # single-rank runs of the two shards from the same (rank 0's = golden) weights
    singles = []
    for r in range(2):
        s = AdvancedTripletReconSegmentationModel(use_gpu=True)
        for k, m in s.model.items():
            m.load_state_dict(golden_sd[k])
        grads = {}
        losses = s.cooperative_step(*W.shard(r, "cuda"), W.CH_MSE, W.SP_CE, do_optim=False,
                                    grad_hook=lambda sol: grads.update({k: m._flat.grad.detach().clone() for k, m in sol.model.items()}))
        singles.append((torch.stack([v.detach().float() for v in losses]).cpu(), grads,
                        {k: (m._bflat.detach().cpu().clone(), m._nbt.detach().cpu().clone()) for k, m in s.model.items()}))
    order = list(singles[0][1].keys())
    summed = torch.cat([(singles[0][1][k] + singles[1][1][k]).cpu() for k in order])
    for r in range(2):
        assert torch.equal(ranks[r]["losses"], singles[r][0])                     # every rank trains on ITS shard
        assert torch.equal(ranks[r]["bucket_sum"], summed)                        # all-reduce(SUM) == g0 + g1, bit for bit
        for k in order:                                                           # BatchNorm statistics stay rank-local (no SyncBN upstream)
            assert torch.equal(ranks[r]["buffers"][k][0], singles[r][2][k][0]) and torch.equal(ranks[r]["buffers"][k][1], singles[r][2][k][1])
    assert any(not torch.equal(ranks[0]["buffers"][k][0], ranks[1]["buffers"][k][0]) for k in order)
    # round 3: the five per-network ranges were exchanged from INSIDE the backward sweep (each as soon as its network's last backward pass
    # had been issued: the decoders and the STN first, the FTN encoder -- the tail of the sweep -- last), in the same order on every rank
    for r in range(2):
        assert sorted(ranks[r]["launched_in_backward"]) == sorted(order), ranks[r]["launched_in_backward"]
        assert ranks[r]["launched_in_backward"][-1] == "image_encoder" and ranks[r]["launched_in_backward"] == ranks[0]["launched_in_backward"]
        assert sorted(ranks[r]["launched_in_backward2"]) == sorted(order)
    # the update: Adam on the MEAN gradient (1/world folded into the kernel), identical on both ranks
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    for k, m in s.model.items():
        m.load_state_dict(golden_sd[k])
        m._flat.grad.copy_(singles[0][1][k] + singles[1][1][k])
        m.mark_grad_written()
    s.grad_scale = 0.5
    s.optimize_all_params()
    for k in order:
        assert torch.equal(ranks[0]["weights"][k], ranks[1]["weights"][k]), k
        assert torch.equal(ranks[0]["weights"][k], s.model[k]._flat_data.cpu()), k
    # per-rank RNG streams: different dropout patterns, still identical weights after the second step
    assert not torch.equal(ranks[0]["drop_masks"]["image"], ranks[1]["drop_masks"]["image"])
    for k in order:
        assert torch.equal(ranks[0]["weights2"][k], ranks[1]["weights2"][k]), k


def test_rccl_world1_smoke(tmp_path):
    """backend "nccl" IS RCCL on ROCm: process-group init, broadcast of the state, one bucket all-reduce inside a real step."""
    _launch(1, "nccl", tmp_path, 29563)
    rec = torch.load(tmp_path / "rank0.pt", weights_only=False)
    assert torch.isfinite(rec["losses"]).all() and torch.isfinite(rec["bucket_sum"]).all()


def test_graph_mode_data_parallel_is_bitwise_the_eager_one(tmp_path):
    """VERDICT r3 item 7.  Two ranks (GPU 0, gloo), three steps with targeted masks and random thresholds: graph-mode DP (forward/backward graph,
    eager all-reduce of the five ranges, Adam graph) ends at bitwise the eager-DP weights and losses on every rank and consumes the same
    host draws.  Then BASELINE configs[3]'s scheme (`mask_type='random'`: per-rank draws, one captured graph per scheme pair): the ranks
    draw different scheme sequences and still hold bit-identical weights after every one of four steps."""
    _launch(2, "gloo", tmp_path, 29565, "graph")
    r = [torch.load(tmp_path / f"graph_rank{k}.pt", weights_only=False) for k in range(2)]
    for k in range(2):
        for a, b in zip(r[k]["eager"]["losses"], r[k]["graph"]["losses"]):
            assert torch.equal(a, b), (k, a, b)
        for name in r[k]["eager"]["weights"]:
            assert torch.equal(r[k]["eager"]["weights"][name], r[k]["graph"]["weights"][name]), (k, name)
        assert r[k]["eager"]["k_next"] == r[k]["graph"]["k_next"]
    for name in r[0]["graph"]["weights"]:
        assert torch.equal(r[0]["graph"]["weights"][name], r[1]["graph"]["weights"][name]), name
    assert not torch.equal(r[0]["graph"]["losses"][0], r[1]["graph"]["losses"][0])          # (each rank trains on its own shard)
    for step in range(4):
        w0, l0 = r[0]["random"]["per_step"][step]
        w1, l1 = r[1]["random"]["per_step"][step]
        assert torch.isfinite(l0).all() and torch.isfinite(l1).all()
        for name in w0:
            assert torch.equal(w0[name], w1[name]), (step, name)
    assert r[0]["random"]["replays"] == r[1]["random"]["replays"] == 4
    assert len(r[0]["random"]["schemes"]) >= 2 or len(r[1]["random"]["schemes"]) >= 2          # several scheme-pair graphs were captured


def test_four_ranks_hold_identical_weights_after_three_steps(tmp_path):
    """the per-network split of the gradient exchange at world size 4 (gloo, all ranks on GPU 0).  With more
    than two ranks the summation order of an element may depend on its position in the exchanged buffer (dist.py), so the assertion is the
    one that matters for training: after each of three steps (fixed-k targeted, dropout, random-k targeted; per-rank shards and RNG
    streams) all four ranks hold bit-identical weights, and the reduced bucket of step 1 is the same on every rank."""
    _launch(4, "gloo", tmp_path, 29567, "world4")
    r = [torch.load(tmp_path / f"w4_rank{k}.pt", weights_only=False) for k in range(4)]
    assert all(x["world"] == 4 for x in r)
    for k in range(1, 4):
        assert torch.equal(r[0]["bucket_sum"], r[k]["bucket_sum"]), k
        assert r[k]["launched_in_backward"] == r[0]["launched_in_backward"] and r[k]["launched_in_backward"][-1] == "image_encoder"
        for step in range(3):
            for name in r[0]["per_step"][step][0]:
                assert torch.equal(r[0]["per_step"][step][0][name], r[k]["per_step"][step][0][name]), (k, step, name)
    assert torch.isfinite(r[0]["bucket_sum"]).all() and float(r[0]["bucket_sum"].abs().max()) > 0
    losses = [x["per_step"][0][1] for x in r]                        # every rank trained on its own shard
    assert not torch.equal(losses[0], losses[1]) and not torch.equal(losses[2], losses[3])
    for step in range(1, 3):                                          # the weights move from step to step
        name = "image_encoder"
        assert not torch.equal(r[0]["per_step"][step][0][name], r[0]["per_step"][step - 1][0][name])
