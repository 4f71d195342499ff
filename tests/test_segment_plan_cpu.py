"""hipgraph.plan_segments (pure host logic of the segment replay): every DAG is covered by linear per-chain segments whose launch order and
event waits imply every edge of the DAG."""
import random

import pytest

from cooperative_training_and_latent_space_data_augmentation_amd.hipgraph import plan_segments


def _check(n, edges):
    chain_of, segs = plan_segments(n, edges)
    seen = [0] * n
    pos, seg_idx = {}, {}
    for si, s in enumerate(segs):
        assert s.nodes, "empty segment"
        for k, v in enumerate(s.nodes):
            seen[v] += 1
            pos[v], seg_idx[v] = k, si
            assert chain_of[v] == s.chain
        for w in s.waits:
            assert any(w is t for t in segs[:si]), "waits for a segment that is launched later"
    assert seen == [1] * n
    # per chain, segments appear in launch order = stream order; "happens before" closure over (stream order, waits)
    last_of_chain = {}
    before = [set() for _ in segs]            # segments known to complete before segment si starts
    for si, s in enumerate(segs):
        if s.chain in last_of_chain:
            p = last_of_chain[s.chain]
            before[si] |= before[p] | {p}
        for w in s.waits:
            wi = next(i for i, t in enumerate(segs) if t is w)
            before[si] |= before[wi] | {wi}
        last_of_chain[s.chain] = si
    for a, b in edges:
        if seg_idx[a] == seg_idx[b]:
            assert pos[a] < pos[b]
        else:
            assert seg_idx[a] in before[seg_idx[b]], f"edge {a}->{b} is not implied by the replay order"
    return chain_of, segs


def test_linear_graph_is_one_segment():
    chain_of, segs = _check(6, [(i, i + 1) for i in range(5)])
    assert len(segs) == 1 and set(chain_of) == {0} and not segs[0].waits


def test_diamond():
    chain_of, segs = _check(8, [(0, 1), (1, 2), (1, 3), (2, 4), (3, 5), (4, 6), (5, 6), (6, 7)])
    assert max(chain_of) == 1
    assert [s.nodes for s in segs] == [[0, 1], [3, 5], [2, 4], [6, 7]]


def test_the_step_shape_two_chains_with_many_forks_and_joins():
    # two chains of 100 nodes, chain B forks from A and joins back every 10 nodes, plus cross edges in the other direction
    edges, n = [], 200
    for i in range(99):
        edges += [(i, i + 1), (100 + i, 100 + i + 1)]
    edges.append((0, 100))
    for k in range(10, 100, 10):
        edges.append((k, 100 + k + 1) if k % 20 else (100 + k, k + 1))
    edges.append((199, 99))
    chain_of, segs = _check(n, edges)
    assert max(chain_of) == 1 and len(segs) <= 24


@pytest.mark.parametrize("seed", range(20))
def test_random_dags(seed):
    rng = random.Random(seed)
    n = rng.randint(5, 60)
    edges = set()
    for b in range(1, n):
        for a in rng.sample(range(b), k=min(b, rng.randint(1, 3))):
            if rng.random() < 0.8 or not any(e[1] == b for e in edges):
                edges.add((a, b))
    _check(n, sorted(edges))


def test_cycle_is_refused():
    from cooperative_training_and_latent_space_data_augmentation_amd._ffi import CtlError
    with pytest.raises(CtlError):
        plan_segments(3, [(0, 1), (1, 2), (2, 1)])
