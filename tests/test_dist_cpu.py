"""N>1 path on CPU: world_size-2 gloo processes exercise the gradient bucket + all-reduce + state broadcast that
bench.py uses with RCCL (host logic only -- the HIP kernels are not involved)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cooperative_training_and_latent_space_data_augmentation_amd import nets
from cooperative_training_and_latent_space_data_augmentation_amd.dist import DataParallel


class _FakeSolver:
    def __init__(self, model):
        self.model, self.grad_scale = model, 1.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                     # deliberately different weights per rank before the broadcast
        model = nets.build_networks(device="cpu")
        solver = _FakeSolver(model)
        before = {k: m._flat_data.clone() for k, m in model.items()}
        dp = DataParallel(solver)
        assert solver.grad_scale == 1.0 / world
        # (1) identical state after broadcast: gather rank 0's checksum
        cs = torch.tensor([float(sum(m._flat_data.double().sum() for m in model.values()))], dtype=torch.float64)
        ref = cs.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(cs, ref)
        if rank != 0:
            assert any(not torch.equal(before[k], m._flat_data) for k, m in model.items())
        # (2) one contiguous bucket: the named .grad tensors are views of it
        total = sum(m._pcount for m in model.values())
        assert dp.bucket.buf.numel() == total
        for m in model.values():
            for n, p in m.named_parameters():
                p.grad.fill_(float(rank + 1))
        assert float(dp.bucket.buf.max()) == rank + 1
        # (3) all-reduce = sum over ranks; the 1/world lives in grad_scale (applied inside the Adam kernel)
        dp.sync_gradients()
        expect = float(sum(range(1, world + 1)))
        for m in model.values():
            for n, p in m.named_parameters():
                assert float(p.grad.min()) == expect and float(p.grad.max()) == expect, n
        assert float((dp.bucket.buf * solver.grad_scale).max()) == expect / world
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_bucket_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
