"""Whole-step HIP graphs: one cooperative-training iteration (both launch chains, forward + backward + 5x Adam, ~1100 kernel
launches) captured once per masking-scheme combination and replayed with ONE host call (the loop body it replaces
is medseg/train_adv_supervised_segmentation_triplet.py:171-237).  The step time then no longer depends on how fast the host can issue
launches, and no allocator runs inside the step.

What may not be baked into a captured launch lives on the device instead:
  * RNG seeds (dropout pattern, soft-mask noise) and the Adam step count: `solver._gstate` (int64[3]), advanced by the `ctl_step_tick`
    launch at the head of the captured step; the dropout / uniform / Adam kernels read it (`ctl_dropout2d_ex`, `ctl_uniform_dev`,
    `ctl_adam_dev`);
  * the random thresholds k of the targeted masks: int32 device scalars the host refreshes before every replay from `np.random`, in the
    reference's draw order (model_util.py:229-230) -- so a seeded run draws the reference's k sequence;
  * the masking scheme of `mask_type='random'` (python `random.shuffle`, model.py:325-329) changes the launch sequence: one graph per
    (image scheme, segmentation scheme) pair, picked per step by the same host draw.
Inputs are copied into static buffers; the returned losses / `solver.z_i` / `last_masks` are the graph's static outputs (valid until
the next replay).  torch.cuda.graph is hipGraph capture of the launches this package enqueues: plumbing, no tracing compiler.

Two ways to replay the captured step (`replay=`): "runtime" = hipGraphLaunch of the captured two-chain graph; "segments" = the same
nodes re-cut into linear per-chain segment graphs launched on two streams (hipgraph.SegmentReplay: the runtime's fast path for linear
graphs; 8 % faster on the bf16 step, whose kernels are too short to hide the general path's per-node cost).  Same results bit for bit."""
from __future__ import annotations

import random
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from ._ffi import CtlError
from .hipgraph import SegmentReplay
from .model_util import _draw_seed

_SCHEMES = ["dropout", "spatial", "channel"]


class _Entry:
    __slots__ = ("graph", "adam_graph", "losses", "k_slots", "masks", "z", "segments")


class CooperativeStepGraph:
    """`step = CooperativeStepGraph(solver, img_cfg, seg_cfg); losses = step(clean, label, noisy)` == `solver.cooperative_step(...)`.

    grad_hook (data parallel): the captured step is split behind `loss.backward()`; the hook (gradient all-reduce) runs eagerly
    between the forward/backward graph and the Adam graph."""

    def __init__(self, solver, img_cfg: Optional[dict], seg_cfg: Optional[dict], separate_training: bool = False,
                 latent_DA: bool = True, grad_hook=None, replay: str = "runtime"):
        if replay not in ("runtime", "segments"):
            raise ValueError("replay must be 'runtime' or 'segments'")
        self.replay_mode = replay
        self.solver, self.img_cfg, self.seg_cfg = solver, img_cfg, seg_cfg
        self.separate_training, self.latent_DA, self.grad_hook = separate_training, latent_DA, grad_hook
        self.entries: Dict[tuple, _Entry] = {}
        self.pool = None
        self.stream = torch.cuda.Stream(device=solver.device)
        self.static_in = None
        dev = solver.device
        self.state = torch.zeros(3, dtype=torch.int64, device=dev)
        self.state[0] = _draw_seed() & (2 ** 62 - 1)                   # respects torch.manual_seed
        self.k_dev = {1: torch.zeros(1, dtype=torch.int32, device=dev), 2: torch.zeros(1, dtype=torch.int32, device=dev)}
        self.replays = 0
        self.k_log = []                # the k values handed to the replays, in draw order (tests compare them with the eager sequence)
        self._dev_adam_count = None    # what state[2] holds on the device after the launches issued so far

    def _new_segments(self, graph):
        other = next((x.segments for x in self.entries.values() if x.segments is not None), None)
        return SegmentReplay(graph, streams_of=other)

    def set_replay_mode(self, mode: str):
        """Switch between the two replay forms of the captured graphs (see the module docstring); "segments" builds the segment graphs
        of every graph captured so far (graphs captured later build theirs at capture)."""
        if mode not in ("runtime", "segments"):
            raise ValueError("replay mode must be 'runtime' or 'segments'")
        if mode == "segments":
            for e in self.entries.values():
                if e.segments is None:
                    e.segments = self._new_segments(e.graph)
        self.replay_mode = mode

    # ------------------------------------------------------------------ host draws, in the reference's order
    def _draw_schemes(self):
        out = []
        for cfg in ((self.img_cfg, self.seg_cfg) if self.latent_DA else ()):
            if cfg is None:
                out.append(None)
            elif cfg["mask_type"] == "random":
                cands = list(_SCHEMES)
                random.shuffle(cands)                                   # model.py:325-329
                out.append(cands[0])
            else:
                out.append(cfg["mask_type"])
        return tuple(out)

    def _draw_ks(self, schemes, zshape):
        """np.random draws of the targeted schemes with random_threshold (model_util.py:229-230 / 291-292), image first."""
        n, c, h, w = zshape
        slot = 0
        for cfg, sc in zip((self.img_cfg, self.seg_cfg), schemes):
            if cfg is None:
                continue
            slot += 1
            if sc in ("channel", "spatial") and cfg["random_threshold"]:
                L = c if sc == "channel" else h * w
                # the value travels as a LAUNCH ARGUMENT of a fill kernel (captured at issue time).  A non-blocking copy from one
                # reused pinned word is read when the copy executes: the host runs several replays ahead of the GPU, so later draws
                # overwrote the word before earlier copies had run and consecutive steps saw the same k
                k = int(L * (np.random.rand() * cfg["max_threshold"]))
                self.k_dev[slot].fill_(k)
                self.k_log.append(k)

    # ------------------------------------------------------------------ capture
    def _adam_step_counts(self):
        return [o.step_count for o in self.solver.optimizers.values()]

    def _run_step(self, schemes, do_optim, hook):
        s = self.solver
        c, l, n = self.static_in
        ov = [({"scheme": sc} if sc is not None else None) for sc in schemes] + [None, None]
        return s.cooperative_step(c, l, n, self.img_cfg, self.seg_cfg, latent_DA=self.latent_DA, separate_training=self.separate_training,
                                  image_override=ov[0], seg_override=ov[1], do_optim=do_optim, grad_hook=hook)

    def _capture(self, schemes) -> _Entry:
        s = self.solver
        counts = self._adam_step_counts()
        if len(set(counts)) != 1:
            raise CtlError("CooperativeStepGraph: the five optimizers must be at the same step count (one device-side counter)")
        # host RNG streams and the BatchNorm buffers are put back after the warm-up: capturing must not advance the training state
        rng = (random.getstate(), np.random.get_state(), torch.get_rng_state())
        saved = {k: (m._bflat.clone(), m._nbt.clone()) for k, m in s.model.items()}
        s.z_i = s.z_s = None                  # tensors the solver keeps hold the previous call's autograd graph alive, and with it the
        s.latent_code = {"image": None, "segmentation": None, "shape": None}      # AccumulateGrad nodes of the flat parameters, bound
        # to the stream of their first use; under capture they would pull the legacy stream into the capture.  The warm-up below
        # re-creates them on the capture stream.
        self.state[2] = counts[0]
        state0 = self.state.clone()
        e = _Entry()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        s._gstate, s._gk = self.state, self.k_dev
        dp = getattr(s, "_dp", None)
        if dp is not None:                    # (data parallel: nothing may start a collective from inside the warm-up step or the capture)
            dp.suspended = True
        try:
            with torch.cuda.stream(self.stream):
                self._run_step(schemes, do_optim=False, hook=None)      # warm-up: plan compilation, lazy tables, per-stream scratch
                for o in s.optimizers.values():                        # (Adam itself needs no warm-up: no lazy state)
                    o.zero_grad()
                for k, m in s.model.items():
                    m._bflat.copy_(saved[k][0])
                    m._nbt.copy_(saved[k][1])
                self.state.copy_(state0)
            self.stream.synchronize()
            s.z_i = s.z_s = None
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            for m in s.model.values():
                m.weights_changed()           # the weight re-pack launches belong INTO the graph (every replay follows an Adam step)
                m._grad_is_zero = False       # ... and so does the step's first gradient fill (every replay follows a step that left gradients)
            e.graph = torch.cuda.CUDAGraph(keep_graph=True)       # (the hipGraph_t stays available to SegmentReplay)
            e.segments = None
            split = self.grad_hook is not None
            with torch.cuda.graph(e.graph, pool=self.pool, stream=self.stream):
                e.losses = self._run_step(schemes, do_optim=not split, hook=None)
            e.adam_graph = None
            if split:
                e.adam_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(e.adam_graph, pool=self.pool, stream=self.stream):
                    s.optimize_all_params()
            e.masks, e.z = dict(s.last_masks), (s.z_i, s.z_s)
            if self.replay_mode == "segments":
                e.segments = self._new_segments(e.graph)
        finally:
            s._gstate = s._gk = None
            if dp is not None:
                dp.suspended = False
            for o, c0 in zip(s.optimizers.values(), counts):           # capturing executed nothing: undo the host-side mirror
                o.step_count = c0
            random.setstate(rng[0])
            np.random.set_state(rng[1])
            torch.set_rng_state(rng[2])
        cur.wait_stream(self.stream)
        return e

    # ------------------------------------------------------------------ one training step
    def __call__(self, clean, label, noisy):
        s = self.solver
        if self.static_in is None:
            self.static_in = tuple(t.detach().clone(memory_format=torch.preserve_format) for t in (clean, label, noisy))
        elif any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(self.static_in, (clean, label, noisy))):
            raise CtlError("CooperativeStepGraph: input shapes / dtypes are fixed at the first call (build one graph object per shape)")
        schemes = self._draw_schemes()
        e = self.entries.get(schemes)
        if e is None:
            e = self.entries[schemes] = self._capture(schemes)
        cur = torch.cuda.current_stream()
        for dst, src in zip(self.static_in, (clean, label, noisy)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        zs = e.z[0].shape if e.z[0] is not None else None
        if zs is not None:
            self._draw_ks(schemes, zs)
        # the Adam step count lives on the device (state[2], advanced by the replay itself).  Anything that stepped the optimizers
        # outside this object since the last replay (an eager cooperative_step for a tail batch of another shape, load_state_dict,
        # bench.py's launch census) moved the host counts but not the device word: re-seed it, as a launch argument, before replaying
        counts = self._adam_step_counts()
        if len(set(counts)) != 1:
            raise CtlError("CooperativeStepGraph: the five optimizers must be at the same step count (one device-side counter)")
        if counts[0] != self._dev_adam_count:
            self.state[2:3].fill_(counts[0])
        if self.replay_mode == "segments":
            if e.segments is None:
                e.segments = self._new_segments(e.graph)
            e.segments.replay()
        else:
            e.graph.replay()
        if e.adam_graph is not None:
            self.grad_hook(s)
            e.adam_graph.replay()
        for o in s.optimizers.values():        # host mirrors of what the replay did on the device
            o.step_count += 1
        self._dev_adam_count = counts[0] + 1
        for m in s.model.values():
            m.weights_changed()
            m.mark_grad_written()              # (the replay left this step's gradients in the buffers: the next zero_grad has work to do)
        s.z_i, s.z_s = e.z
        s.last_masks = dict(e.masks)
        self.replays += 1
        return e.losses
