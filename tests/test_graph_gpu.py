"""Whole-step HIP graphs (graph.py): a replayed step must be THE SAME computation as the eager step -- every kernel is
deterministic, so with deterministic masks the weights, BatchNorm buffers and losses are compared bit for bit."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd import ops  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph  # noqa: E402
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel  # noqa: E402

DEV = "cuda"
CH_MSE = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
SP_CE = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
CH_MSE_RT = dict(CH_MSE, random_threshold=True)
SP_CE_RT = dict(SP_CE, random_threshold=True)
RAND_MSE = {"loss_name": "mse", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": False}
RAND_CE = {"loss_name": "ce", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": False}
DROP_MSE = {"loss_name": "mse", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
DROP_CE = {"loss_name": "ce", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}


def dev(x):
    x = x.to(DEV)
    return x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x.contiguous()


def _solver(golden_sd):
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    for k, m in s.model.items():
        m.load_state_dict(golden_sd[k])
    return s


def _state(s):
    torch.cuda.synchronize()
    return ({k: m._flat_data.detach().cpu().clone() for k, m in s.model.items()},
            {k: (m._bflat.detach().cpu().clone(), m._nbt.detach().cpu().clone()) for k, m in s.model.items()},
            {k: (o.exp_avg.cpu().clone(), o.exp_avg_sq.cpu().clone(), o.step_count) for k, o in s.optimizers.items()})


def _same(a, b):
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), f"weights of {k}"
        assert torch.equal(a[1][k][0], b[1][k][0]) and torch.equal(a[1][k][1], b[1][k][1]), f"BatchNorm buffers of {k}"
        assert torch.equal(a[2][k][0], b[2][k][0]) and torch.equal(a[2][k][1], b[2][k][1]) and a[2][k][2] == b[2][k][2], f"Adam state of {k}"


@pytest.mark.parametrize("replay", ["runtime", "segments"])
@pytest.mark.parametrize("cfgs", [(CH_MSE, SP_CE), (CH_MSE_RT, SP_CE_RT)], ids=["fixed_k", "random_k"])
@pytest.mark.parametrize("two_streams", [True, False])
def test_graph_replay_is_bitwise_the_eager_step(golden_sd, cfgs, two_streams, replay):
    """4 training steps, eager vs graph replay, from the same weights and the same seeded host RNG (the random thresholds k are drawn
    by the host per replay in the reference's order and handed over in device memory).  replay = "segments": the captured graph re-cut
    into linear per-chain segment graphs on two streams (hipgraph.SegmentReplay) -- same nodes, same edges, same bits."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(4, 64, 64, seed=5))
    res = []
    for use_graph in (False, True):
        s = _solver(golden_sd)
        s.two_streams = two_streams
        np.random.seed(3)
        g = CooperativeStepGraph(s, *cfgs, replay=replay) if use_graph else None
        losses = []
        for _ in range(4):
            l = g(clean, label, noisy) if use_graph else s.cooperative_step(clean, label, noisy, *cfgs)
            losses.append(torch.stack([v.detach().float() for v in l]).cpu())
        res.append((losses, _state(s), np.random.rand()))
        if use_graph:
            assert g.replays == 4 and len(g.entries) == 1
            if replay == "segments":
                d = next(iter(g.entries.values())).segments.describe()
                assert d["chains"] == (2 if two_streams else 1) and d["segments"] >= d["chains"] and d["nodes"] > 300, d
                assert (d["events"] > 0) == two_streams, d
                if two_streams:          # the second chain's stream was probed against the launch stream (another hardware queue)
                    assert d["queue_probe"] and all(q["overlap"] for q in d["queue_probe"]), d
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b), (a, b)
    _same(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]                       # both consumed the same number of np.random draws


def test_graph_capture_does_not_advance_training_state(golden_sd):
    """Capturing (warm-up + capture) must leave weights, BatchNorm buffers, Adam state and the host RNG streams untouched; eager calls
    mixed with replays keep working (weights are re-packed after a replay)."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(2, 64, 64, seed=6))
    s = _solver(golden_sd)
    s.cooperative_step(clean, label, noisy, CH_MSE, SP_CE)              # one eager step first: non-trivial Adam state
    before = _state(s)
    g = CooperativeStepGraph(s, CH_MSE, SP_CE)           # (draws the device RNG's seed from torch's host generator)
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    probe = (random.random(), np.random.rand(), float(torch.rand(1)))
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    g.static_in = (clean, label, noisy)
    e = g._capture(g._draw_schemes())
    _same(before, _state(s))
    assert probe == (random.random(), np.random.rand(), float(torch.rand(1)))
    g.entries[(CH_MSE["mask_type"], SP_CE["mask_type"])] = e
    ref = _solver(golden_sd)
    ref.cooperative_step(clean, label, noisy, CH_MSE, SP_CE)
    for _ in range(2):
        g(clean, label, noisy)
        ref.cooperative_step(clean, label, noisy, CH_MSE, SP_CE)
    assert torch.equal(s.predict(noisy), ref.predict(noisy))           # eager inference after replays sees the updated weights
    la, lb = s.cooperative_step(clean, label, noisy, CH_MSE, SP_CE), ref.cooperative_step(clean, label, noisy, CH_MSE, SP_CE)
    assert all(torch.equal(u.detach(), v.detach()) for u, v in zip(la, lb))
    _same(_state(ref), _state(s))


def test_graph_dropout_draws_a_new_pattern_every_replay(golden_sd):
    """Dropout masks under replay: the pattern comes from the device-resident RNG state, so it changes from replay to replay and
    keeps the Bernoulli(0.5) statistics; the Adam step count advances on the device."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(4, 64, 64, seed=7))
    s = _solver(golden_sd)
    g = CooperativeStepGraph(s, DROP_MSE, DROP_CE)
    pats = []
    for _ in range(4):
        losses = g(clean, label, noisy)
        assert all(torch.isfinite(v) for v in losses)
        pats.append((s.last_masks["image"].clone(), s.last_masks["seg"].clone()))
    for a, b in zip(pats, pats[1:]):
        assert not torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])
    assert not torch.equal(pats[0][0], pats[0][1])                      # image and segmentation codes get different patterns
    frac = torch.cat([p[0].flatten() for p in pats]).mean().item()
    assert 0.4 < frac < 0.6
    assert int(g.state[2]) == 4 and all(o.step_count == 4 for o in s.optimizers.values())
    s.cooperative_step(clean, label, noisy, DROP_MSE, DROP_CE)           # the eager path continues from the replayed state
    assert all(o.step_count == 5 for o in s.optimizers.values())


def test_graph_random_scheme_follows_the_seeded_host_draws(golden_sd):
    """mask_type='random': one graph per (image scheme, segmentation scheme); the scheme sequence is the eager path's for the same seed."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(2, 64, 64, seed=8))
    seqs = []
    for use_graph in (False, True):
        s = _solver(golden_sd)
        random.seed(4); np.random.seed(4)
        g = CooperativeStepGraph(s, RAND_MSE, RAND_CE) if use_graph else None
        seq = []
        for _ in range(6):
            if use_graph:
                sch = g._draw_schemes
                drawn = []
                g._draw_schemes = lambda: drawn.append(sch()) or drawn[-1]
                losses = g(clean, label, noisy)
                g._draw_schemes = sch
                seq.append(drawn[0])
            else:
                got = []
                orig = s.perturb_latent_code
                s.perturb_latent_code = lambda *a, **kw: (lambda r: (got.append(s.last_scheme), r)[1])(orig(*a, **kw))
                losses = s.cooperative_step(clean, label, noisy, RAND_MSE, RAND_CE)
                s.perturb_latent_code = orig
                seq.append(tuple(got))
            assert all(torch.isfinite(v) for v in losses)
        seqs.append(seq)
        if use_graph:
            assert len(g.entries) == len(set(seq)) >= 2
    assert seqs[0] == seqs[1]


def test_dropout_mask_has_the_reference_semantics(golden_sd):
    """perturb_latent_code(..., 'dropout') returns upstream's mask (model.py:334-336): 1 where the dropped-out code EQUALS the input."""
    s = _solver(golden_sd)
    z = torch.relu(torch.randn(3, 128, 4, 4, generator=torch.Generator().manual_seed(2)))      # ReLU codes: exact zeros exist
    keep = (torch.rand(3, 128, generator=torch.Generator().manual_seed(3)) > 0.5).float()
    masked, mask = s.perturb_latent_code(dev(z), s.model["image_decoder"], perturb_type="dropout", threshold=0.5, if_detach=True,
                                         override={"keep": keep.to(DEV)})
    ref_masked, ref_mask = O.dropout2d_with_keep(z, 0.5, keep)
    assert torch.equal(masked.cpu(), ref_masked) and torch.equal(mask.cpu(), ref_mask) and mask.shape == z.shape
    out, kp, m2 = ops.dropout2d(dev(z), 0.5, seed=9, want_mask=True)
    assert torch.equal(m2.cpu(), (out.cpu() == z).float())


def test_unsynchronised_replays_see_their_own_k(golden_sd):
    """the host runs several replays ahead of the GPU; every replay must still mask with ITS k (the reference's np.random
    sequence, model_util.py:229-230), not with a later draw.  No host sync between the replays; the per-step masks are cloned
    stream-ordered and counted afterwards."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(4, 64, 64, seed=5))
    s = _solver(golden_sd)
    g = CooperativeStepGraph(s, CH_MSE_RT, SP_CE_RT)
    g(clean, label, noisy)                               # capture + first replay
    torch.cuda.synchronize()
    np.random.seed(11)
    expect = []
    rs = np.random.RandomState(11)
    for _ in range(12):
        expect.append((int(128 * (rs.rand() * 0.5)), int(16 * (rs.rand() * 0.5))))      # z is 128 x 4 x 4 at 64^2: L = 128 / 16
    del g.k_log[:]
    masks = []
    for _ in range(12):
        g(clean, label, noisy)
        masks.append((s.last_masks["image"].clone(), s.last_masks["seg"].clone()))   # stream-ordered copies, no host sync
    torch.cuda.synchronize()
    assert g.k_log == [k for pair in expect for k in pair]
    assert len(set(expect)) > 6                             # the draws do differ from step to step
    for (mi, ms), (ki, ks) in zip(masks, expect):
        assert ((mi == 0).flatten(1).sum(1) == ki).all(), (ki, (mi == 0).flatten(1).sum(1))
        assert ((ms == 0).flatten(1).sum(1) == ks).all(), (ks, (ms == 0).flatten(1).sum(1))


def test_eager_steps_between_replays_keep_the_adam_count(golden_sd):
    """graph, eager, graph == three eager steps, bit for bit (the device-side Adam step counter is re-seeded when the
    optimizers were stepped outside the graph object)."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(2, 64, 64, seed=6))
    ref = _solver(golden_sd)
    for _ in range(4):
        ref.cooperative_step(clean, label, noisy, CH_MSE, SP_CE)
    s = _solver(golden_sd)
    g = CooperativeStepGraph(s, CH_MSE, SP_CE)
    g(clean, label, noisy)
    s.cooperative_step(clean, label, noisy, CH_MSE, SP_CE)
    g(clean, label, noisy)
    g(clean, label, noisy)
    assert int(g.state[2]) == 4
    _same(_state(ref), _state(s))


def test_replay_mode_can_be_switched_between_replays(golden_sd):
    """runtime replay, segment replay, runtime replay ... of ONE captured graph continue one trajectory: bitwise the all-runtime run."""
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(4, 64, 64, seed=9))
    res = []
    for modes in (["runtime"] * 4, ["runtime", "segments", "segments", "runtime"]):
        s = _solver(golden_sd)
        g = CooperativeStepGraph(s, CH_MSE, SP_CE)
        losses = []
        for m in modes:
            g.set_replay_mode(m)
            losses.append(torch.stack([v.detach().float() for v in g(clean, label, noisy)]).cpu())
        res.append((losses, _state(s)))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b), (a, b)
    _same(res[0][1], res[1][1])


def test_segment_replays_move_off_the_launch_streams_hardware_queue(golden_sd):
    """HIP maps streams onto a few hardware queues in creation order; a second chain on the launch stream's queue overlaps nothing.  Six
    SegmentReplay objects in a row (each takes the next stream of torch's pool): every one ends up on a stream that overlaps."""
    from cooperative_training_and_latent_space_data_augmentation_amd.hipgraph import SegmentReplay
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(2, 64, 64, seed=10))
    s = _solver(golden_sd)
    g = CooperativeStepGraph(s, CH_MSE, SP_CE)
    g(clean, label, noisy)
    e = next(iter(g.entries.values()))
    g.replay_mode = "segments"
    tried = 0
    for _ in range(6):
        e.segments = SegmentReplay(e.graph)
        losses = g(clean, label, noisy)
        assert all(torch.isfinite(v) for v in losses)
        q = e.segments.describe()["queue_probe"]
        assert q and q[0]["overlap"], q
        tried += q[0]["streams_tried"]
    assert tried >= 6


def test_probed_side_streams_are_kept_per_launch_stream(golden_sd, monkeypatch):
    """the queue probe of launch stream B must not replace the side stream that launch stream A was probed against (the probe used to
    mutate one shared list and A was never re-probed).  With a probe that reports a collision for B's first candidate only: A keeps its stream,
    B gets another one, and a replay on either launch stream uses its own."""
    from cooperative_training_and_latent_space_data_augmentation_amd import hipgraph
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(2, 64, 64, seed=10))
    s = _solver(golden_sd)
    g = CooperativeStepGraph(s, CH_MSE, SP_CE, replay="segments")
    g(clean, label, noisy)                                   # capture + first replay: the real probe for the current launch stream
    seg = next(iter(g.entries.values())).segments
    a = torch.cuda.current_stream()
    side_a = list(seg._pool["side_of"][a.cuda_stream])
    calls = []

    def fake_ratio(cur, side, us=300):
        calls.append((cur.cuda_stream, side.cuda_stream))
        return 2.0 if len(calls) == 1 else 1.0               # "collision" for the first candidate, overlap for its replacement
    monkeypatch.setattr(hipgraph, "streams_overlap_ratio", fake_ratio)
    b = torch.cuda.Stream()
    seg.prepare(b)
    side_b = seg._pool["side_of"][b.cuda_stream]
    assert len(calls) == 2 and calls[0][0] == b.cuda_stream
    assert [x.cuda_stream for x in seg._pool["side_of"][a.cuda_stream]] == [x.cuda_stream for x in side_a], "A's probed stream was replaced"
    assert side_b[0].cuda_stream != side_a[0].cuda_stream
    monkeypatch.undo()
    la = torch.stack([v.detach().float() for v in g(clean, label, noisy)]).cpu()
    assert torch.isfinite(la).all()
