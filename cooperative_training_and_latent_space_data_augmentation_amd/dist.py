"""Single-node data parallelism for the cooperative step: one process per GPU, `torch.distributed` ("nccl" == RCCL on
ROCm, xGMI between the 8 GPUs of a node; "gloo" on CPU for tests).

The reference has no distributed code.  The path shards naturally over samples: BatchNorm statistics
stay rank-local (no SyncBN upstream => DDP semantics), the saliency top-k is per image, so the ONLY exchange is the
gradient.  All five networks' flat gradients live in one contiguous bucket (2.53 M fp32 = 10.1 MB); the 1/world_size is
folded into the Adam kernel (`grad_scale`).

Round 3 (SURVEY 5 comm row / 8(e): "a single bucket, or 5 per-module buckets overlapped with the tail of backward"): the bucket is
exchanged as its five per-network ranges.  A network's range is all-reduced (async, on the communicator's stream) as soon as the LAST
backward pass of that network has been issued -- the decoders and the STN finish well before the FTN encoder, whose backward is the
tail of the sweep -- and each network's Adam launch waits for its own range only.  With 17.7 MB per GPU on the ring the transfer is
latency- not bandwidth-bound, so the point of the split is the overlap, not the size.  Every rank ends with the same bits in every
range.  Against the single-bucket exchange the sums are bit-identical at world size 2 (one addition per element: tested,
tests/test_dist_gpu.py); with more ranks a ring all-reduce ties each element's summation order to its position in the buffer, so
splitting the bucket may change the fp32 rounding of a sum -- identical across ranks, not necessarily against the single bucket.
No scaling curve has been measured by the builder (1-GPU boxes); see DESIGN.md section 5."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, nets: Dict[str, "torch.nn.Module"]):
        total = sum(n._pcount for n in nets.values())
        first = next(iter(nets.values()))
        self.buf = torch.zeros(total, dtype=torch.float32, device=first._flat_data.device)
        self.ranges: Dict[str, torch.Tensor] = {}
        off = 0
        for name, n in nets.items():
            self.ranges[name] = self.buf[off:off + n._pcount]
            n.bind_grad_buffer(self.ranges[name])
            off += n._pcount


class DataParallel:
    """Wraps a solver: identical start on every rank, the gradient exchange of every step.

    `sync_gradients(solver)` is the `grad_hook` of `cooperative_step` (called between backward and the optimizer): it launches the
    all-reduce of every range that was not already launched from inside the backward sweep and, with `wait=True` (default), makes the
    current stream wait for all of them -- the blocking single-exchange semantics of rounds 1-2.  `overlap=True` additionally arms the
    per-network launch from inside backward; with `wait=False` as the hook the solver's `optimize_all_params` waits per network."""

    def __init__(self, solver, process_group=None, overlap: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.solver, self.pg = solver, process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket = GradBucket(solver.model)
        self._works: Dict[str, object] = {}
        self.launched_in_backward = []          # (diagnostics / tests) names, in launch order, of the last step
        self.suspended = False                  # graph capture / its warm-up step: no exchange from inside backward
        self.armed = False                      # set per step by the solver: True only if that step was given a grad_hook
        solver.grad_scale = 1.0 / self.world
        solver._dp = self
        self.overlap = overlap
        for name, net in solver.model.items():
            net._on_grads_complete = (lambda n=name: self._from_backward(n)) if overlap else None
        self.broadcast_state()

    def broadcast_state(self, src: int = 0):
        for net in self.solver.model.values():
            dist.broadcast(net._flat_data, src, group=self.pg)
            dist.broadcast(net._bflat, src, group=self.pg)
            dist.broadcast(net._nbt, src, group=self.pg)
            net.weights_changed()

    # ------------------------------------------------------------------ exchange
    def _launch(self, name: str):
        if name not in self._works:
            # (NCCL/RCCL: the collective is enqueued on the communicator's stream behind everything issued so far on the CURRENT stream,
            #  which is the stream that produced this range; gloo: host-side, the call returns a handle as well)
            self._works[name] = dist.all_reduce(self.bucket.ranges[name], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def begin_step(self, exchange: bool):
        """Called by `cooperative_step` at the head of every step.  Handles left over by a step that ended without its waits
        (`do_optim=False`, an exception between launch and wait) are waited for and dropped: a stale handle would make `_launch`
        skip that network's all-reduce in THIS step.  The launch from inside backward is armed only if the step has a grad_hook:
        `grad_hook=None` means 'no exchange'."""
        for name in list(self._works):
            self.wait(name)
        self.launched_in_backward = []
        self.armed = bool(exchange)

    def _from_backward(self, name: str):
        if not self.armed:
            return
        if self.suspended or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return                                # (graph mode: the exchange runs eagerly between the two graphs)
        self.solver.model[name].collect_deferred_grads()      # the parked per-pass gradients of this network -> its range of the bucket
        self.launched_in_backward.append(name)
        self._launch(name)

    def wait(self, name: str):
        w = self._works.pop(name, None)
        if w is not None:
            w.wait()

    def sync_gradients(self, solver=None, wait: bool = True):
        for name in self.bucket.ranges:
            self._launch(name)
        if wait:
            for name in list(self._works):
                self.wait(name)

    def launch_remaining(self, solver=None):
        """grad_hook form without the waits: `optimize_all_params` waits for each network's range in front of its Adam launch."""
        self.sync_gradients(solver, wait=False)
