"""Single-node data parallelism for the cooperative step: one process per GPU, `torch.distributed` ("nccl" == RCCL on
ROCm, xGMI between the 8 GPUs of a node; "gloo" on CPU for tests).

The reference has no distributed code.  The path shards naturally over samples: BatchNorm statistics
stay rank-local (no SyncBN upstream => DDP semantics), the saliency top-k is per image, so the ONLY exchange is the
gradient: all five networks' flat gradients live in one contiguous bucket (2.53 M fp32 = 10.1 MB) that is all-reduced
once per step after `loss.backward()`; the 1/world_size is folded into the Adam kernel (`grad_scale`).  With 17.7 MB per
GPU on the ring the transfer is latency- not bandwidth-bound, hence one bucket rather than per-layer buckets."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, nets: Dict[str, "torch.nn.Module"]):
        nets = list(nets.values())
        total = sum(n._pcount for n in nets)
        self.buf = torch.zeros(total, dtype=torch.float32, device=nets[0]._flat_data.device)
        off = 0
        for n in nets:
            n.bind_grad_buffer(self.buf[off:off + n._pcount])
            off += n._pcount


class DataParallel:
    """Wraps a solver: identical start on every rank, one gradient all-reduce per step."""

    def __init__(self, solver, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.solver, self.pg = solver, process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket = GradBucket(solver.model)
        solver.grad_scale = 1.0 / self.world
        self.broadcast_state()

    def broadcast_state(self, src: int = 0):
        for net in self.solver.model.values():
            dist.broadcast(net._flat_data, src, group=self.pg)
            dist.broadcast(net._bflat, src, group=self.pg)
            dist.broadcast(net._nbt, src, group=self.pg)
            net.weights_changed()

    def sync_gradients(self, solver=None):
        dist.all_reduce(self.bucket.buf, op=dist.ReduceOp.SUM, group=self.pg)
