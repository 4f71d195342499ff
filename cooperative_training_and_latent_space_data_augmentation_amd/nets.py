"""FTN/STN networks of the reference, re-designed for MI355X: each network pass is ONE C call (`ctl_plan_run`) over a
pre-compiled list of kernel launches ("plan") working on NHWC fp32 tensors.

Mirrors (names + state_dict keys, so upstream `.pth` files load unchanged):
    MyEncoder            medseg/models/ebm/encoder_decoder.py:351-415
    MyDecoder            medseg/models/ebm/encoder_decoder.py:418-453
    Dual_Branch_Encoder  medseg/models/ebm/encoder_decoder.py:456-503
    res_convdown / res_up_family blocks: :19-68 / :285-348

HBM data layout
    parameters   one flat fp32 buffer per network (`_flat`), every tensor 256-B aligned; the named nn.Parameters are
                 OIHW views into it (checkpoint-compatible); gradients likewise (`_flat.grad`); Adam runs on the flat
                 buffer in one launch.
    weights      re-packed once per optimizer step into MFMA-fragment order (forward + dgrad variants), `_wp`.
    activations  NHWC; raw (pre-BatchNorm) conv outputs are what is stored.  BatchNorm-apply + LeakyReLU is never
                 materialised on the main path: the consumer conv applies it while staging its input tile, the
                 residual tail is fused into the 1x1 `conv_input` epilogue, nearest-upsampling is pure indexing.

BatchNorm modes (SURVEY 8a row 4 / model_util.py:414-451): "A" train+track, "B" train without running-stat update and
without gamma/beta gradients, "C" eval (running statistics).
"""
from __future__ import annotations

import ctypes
from collections import namedtuple
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _ffi, init as _init
from ._ffi import lib, check, OP_DTYPE

(S_X, S_P, S_B, S_NBT, S_WP, S_ACT, S_SCR, S_OUT0, S_OUT1, S_DOUT0, S_DOUT1, S_GRAD, S_DX, S_BSCR, S_TAB, S_STATE, S_KEEP) = range(17)
N_SLOTS = 17
SLOPE = 0.2
EPS = 1e-5
MOMENTUM = 0.1
# Plan-compiler switches (module attributes, no environment reads; tools/ab.py flips them for A/B runs)
SMALL_CIN = True         # K-packed taps for the Cin <= 4 first layers
PHASE_CONVS = True       # 2x2 phase forms of the up-sampled forward / stride-2 data gradient
FUSE_BNBWD = True        # fp32: the in-block BatchNorm-backward reduction inside the data-gradient epilogue (its own instantiation, EPI 3: no scratch).
                         # Alone it was worth nothing in rounds 2-3 (matrix-bound host kernel); together with the staged apply it removes the last
                         # element-wise BatchNorm-backward passes of the fp32 step: 955 -> 875 launches, 17.43 -> 17.26 ms same-box
FUSE_BNAPPLY = True      # fp32: the BatchNorm-backward apply passes inside the staging of their consumers (see FUSE_BNAPPLY16); the tail's alone: 40
                         # launches and a tensor pass less per step for 17.52 -> 17.45 ms; the in-block one follows FUSE_BNBWD
FUSE_BNBWD16 = True      # the same in the bf16 family, where the data gradient is not matrix-bound
FUSE_XOUT16 = True       # bf16: the data-gradient conv that stages a virtual tensor also WRITES it (its tiles' interiors), and the weight gradient of the
                         # same layer -- issued behind it -- reads it as a plain tensor instead of evaluating it again in every cin-chunk block
                         # (the virtual output gradient costs the bf16 weight gradients +14-44 % in isolation): 9.76 -> 9.49 ms same-box
FUSE_XOUT = False        # fp32: the same, parity-tested, and 16.64 -> 16.85 ms: the extra stores sit in the serial staging phase of matrix-bound kernels
FUSE_POOL = True         # both families: the tail epilogue of a 1x1 host also writes sumpool2(g) for a consuming nearest-upsample block (no ctl_sumpool2 pass)
FUSE_PAIR = True         # fp32: the same for the fp32 family (these pairs are the 256^2 x 16-channel tensors: 0.87 ms of reduce + apply passes per step)
FUSE_PAIR16 = True       # bf16: the conv-BatchNorm-activation pairs at the head of the encoders get g = dAct * act' and their sums from the launch that
                         # writes dAct (CTL_EPI_BNBWD on it), and no apply pass either (FUSE_BNAPPLY16)
FUSE_TAIL16 = True       # bf16: CTL_EPI_TAILBWD in the bf16 family (the tail's reduction pass and the separately rounded dOut disappear)
FUSE_BNAPPLY16 = True    # bf16: the BatchNorm-backward apply passes of the residual blocks run inside the staging of their consumers (pro_affine 2 / dy2)
X3 = True                # fp32: the 2x2 / 3x3 / 4x4 convs with cin, cout multiples of 16 contract on the bf16 matrix pipe over an exact three-way
                         # bf16 split of both fp32 operands (CTL_DT_X3, csrc/ctl_conv_x3_stage.h): same tensors, same epilogues, error per product
                         # <= 2^-24 (1 + 2^-8) (one fp32 multiply's rounding; typically 2^-25), 16/6 of the fp32 MFMA rate.  A MODULE-level
                         # switch read when a CtlNet is built: set nets.X3 = False BEFORE constructing the solver (bench.py --set does)
X3_WGRAD = True          # ... and so do the weight gradients of those layers (3x3 stride 1 / 2, 2x2 stride 2)
GROUP_WGRAD = True       # fp32 / X3: the weight gradients of a backward plan that can share a launch (ctl_wgrad_group_class) are DEFERRED to the end of
                         # the plan and served in groups of up to 8 by one launch each (CTL_OP_WGRAD_GROUP): nothing in the plan reads dW, and per
                         # launch ~15 of 40 us are fixed.  Safe to reorder: plan arenas are bump-allocated (no tensor is reused inside a plan) and
                         # every in-place pass on a gradient tensor (apply, accumulate epilogues) is emitted BEFORE the weight gradient that reads it
GROUP_WGRAD_BF16 = True  # bf16 family: the weight gradients of one kernel instantiation are STACKED in one launch (up to 8 members along blockIdx.x): the bf16
                         # step is launch-bound.  A member's pixel splits are dealt in proportion to its work (ctl_wgrad_group_plan; never more than a launch of
                         # its own takes), so its partial sums are added in another order than the single launch's -- equal bit for bit only when the split
                         # count equals ctl_wgrad_splits' (tests/test_bf16_gpu.py compares both: exact with own splits, 1e-5 with the proportional ones)
FUSE_TAIL = True         # fp32: the residual tail's BatchNorm-backward reduction inside the launch that writes dOut (CTL_EPI_TAILBWD): dOut is
                         # never materialised, the stand-alone reduction launch (read dOut, out, v; write dS) disappears
CONV_WORDS = _ffi.CONV_DTYPE.itemsize // 4      # ctl_conv as int32 words at the head of ctl_op.i
ALIGN_F = 64            # floats (256 B)

T = namedtuple("T", "ref n h w c b16", defaults=(False,))     # tensor descriptor: ref = (slot, byte offset), NHWC dims, bf16 storage?


def _rup(x: int, a: int) -> int:
    return (x + a - 1) // a * a


class _Holder(nn.Module):
    """Plain container (stands in for nn.Sequential / the block classes; it owns no compute)."""


class _ConvP(nn.Module):
    def __init__(self, cin, cout, ks, transposed=False):
        super().__init__()
        self.cin, self.cout, self.ks, self.transposed = cin, cout, ks, transposed
        shape = (cin, cout, ks, ks) if transposed else (cout, cin, ks, ks)
        self.weight = nn.Parameter(torch.zeros(shape))
        self.bias = nn.Parameter(torch.zeros(cout))


class _BNP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class ConvInfo:
    __slots__ = ("key", "cin", "cout", "ks", "transposed", "w_off", "b_off", "wp_fwd", "wp_dgrad", "wp_sub", "wp_up", "wp_upf", "wp_s2d", "wp_ph", "wp_c4")


class BNInfo:
    __slots__ = ("key", "c", "g_off", "b_off", "rm_off", "rv_off", "nbt_idx")


class Arena:
    """Bump allocator of one plan workspace.  `b16`: the tensors living here are network-internal and stored as bf16 (BASELINE config 3);
    network inputs / outputs (the external slots) are fp32 in every configuration."""

    def __init__(self, slot, b16=False, pp=(0, 1)):
        # pp = (p, P): this plan is pass p of P passes of one network that share the arena in the N-STACKED layout: every tensor is
        # allocated for P * n images and pass p works on images [p * n, (p + 1) * n), every per-group coefficient table for P * groups
        # rows with this pass' rows at p * groups -- so that ONE backward launch chain can later treat the P passes as one batch of
        # P * n images in P * groups BatchNorm groups (round 6, CtlNet.run_backward_stacked)
        self.slot, self.size, self.b16, self.pp = slot, 0, b16, pp

    def alloc(self, nbytes: int):
        off = self.size
        self.size += _rup(max(int(nbytes), 4), 256)
        return (self.slot, off)

    def tensor(self, n, h, w, c, *_, b16=None) -> T:
        b16 = self.b16 if b16 is None else b16
        p, P = self.pp
        one = (2 if b16 else 4) * n * h * w * c
        slot, off = self.alloc(P * one)
        return T((slot, off + p * one), n, h, w, c, b16)

    def rows(self, floats: int, groups: int):
        """A per-group table [groups][floats] of fp32 (BatchNorm coefficients / saved statistics)."""
        p, P = self.pp
        one = 4 * floats * groups
        slot, off = self.alloc(P * one)
        return (slot, off + p * one)


class _ArenaLease:
    """One activation / backward arena on loan from a network's pool; goes back to the free list when the last holder (the autograd
    context of the pass) dies."""
    __slots__ = ("free", "t")

    def __init__(self, free, t):
        self.free, self.t = free, t

    def __del__(self):
        if self.free is not None:
            self.free.append(self.t)


class ArenaPool:
    """Plan workspaces (saved activations: 289 MB for a bs16 256^2 decoder pass) are recycled per (stream, size) instead of going
    through the caching allocator every pass: with two launch chains its block reuse depends on event timing, and a device
    allocation inside a training step is a stall (bench.py counts them).  Reuse is safe without events: an arena is only handed out
    again on the stream it was last used on (forward and backward of a pass run on one stream), i.e. in launch order.  Under HIP-graph
    capture the pool is bypassed (the graph's private pool owns that memory)."""

    def __init__(self):
        self._free: Dict[tuple, list] = {}

    def acquire(self, nbytes: int, device) -> "_ArenaLease":
        nbytes = max(int(nbytes), 256)
        if torch.cuda.is_current_stream_capturing():
            return _ArenaLease(None, torch.empty(nbytes, dtype=torch.uint8, device=device))
        free = self._free.setdefault((torch.cuda.current_stream().cuda_stream, nbytes), [])
        t = free.pop() if free else torch.empty(nbytes, dtype=torch.uint8, device=device)
        return _ArenaLease(free, t)

    def clear(self):
        for lst in self._free.values():
            lst.clear()


class Plan:
    __slots__ = ("ops", "n_ops", "act_bytes", "scr_bytes", "bscr_bytes", "rec", "out_shapes", "table_np", "table_dev", "groups", "bn_log",
                 "replay", "main_ops", "tail_ops")


# plan ops that only PRODUCE parameter gradients (nothing in a backward plan reads them): a split plan runs them as its tail, on another
# stream than the data-gradient chain (CtlNet._run_backward_stacked)
_TAIL_KINDS = (_ffi.OP_WGRAD, _ffi.OP_WGRAD_GROUP, _ffi.OP_WGRAD_REDUCE, _ffi.OP_WGRAD_REDUCE_BATCH, _ffi.OP_CHAN_SUM_FINALIZE)


def _stack_rec(obj, P: int):
    """The forward record of pass slot 0 seen as ONE pass over the P stacked batches: every tensor descriptor grows to P * n images (its
    reference is slot 0's = the base of the stacked tensor), the per-group coefficient tables keep their base (P * groups rows)."""
    if isinstance(obj, T):
        return obj._replace(n=obj.n * P)
    if isinstance(obj, dict):
        return {k: _stack_rec(v, P) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_stack_rec(v, P) for v in obj)
    return obj


class _StackedForward:
    """What a backward compiler reads of a forward plan (`rec`, `groups`), for the stacked view of P pass slots."""
    __slots__ = ("rec", "groups")

    def __init__(self, rec, groups):
        self.rec, self.groups = rec, groups


class PassStack:
    """The passes of ONE network in one training step whose backward runs as one launch chain (round 6).

    The loss of a step is `standard + hard` and the hard branch's inputs are detached (train.py:216-230, model.py:502-518, 547): the
    standard and the hard-example pass of a network are independent given their inputs, share the weights, and their backward sweeps
    have the same shape.  Their FORWARD passes cannot be batched (the hard input is generated from the standard pass' codes), so each
    forward pass p writes its activations into slot p of one arena laid out as if a single pass had run on the P * n stacked images
    (Arena.pp); the backward then IS a single pass over P * n images in P * groups BatchNorm groups -- one data-gradient chain, one
    weight-gradient contraction over both passes (no second set of split-K partials, no per-pass accumulation), half the launches.
    A pass in BatchNorm mode B (frozen gamma / beta) is a group that adds nothing to their gradients (ctl_bn_bwd_finalize_ex)."""

    def __init__(self, net: "CtlNet", P: int = 2):
        self.net, self.P = net, P
        self.lease = None                 # the stacked activation arena
        self.outs = None                  # stacked outputs, [P * n, c, h, w] each; pass p returns the views [p * n, (p + 1) * n)
        self.shape = None                 # (n, h, w, groups) of the first pass: later passes must agree or run unstacked
        self.plans, self.modes, self.xs, self.roots, self.need_dx, self.seqs = [], [], [], [], [], []
        self.douts = []                   # per pass: list over outputs of the gradient received so far (None = none yet)
        self.keep = []                    # leases / tensors that must outlive the launches issued from them on other streams
        self.events = []                  # per pass: events behind which the received gradients are complete
        self.done = False                 # the backward has been issued

    @property
    def filled(self) -> int:
        return len(self.plans)

    def accepts(self, n, h, w, groups) -> bool:
        return (not self.done and self.filled < self.P and self.net.drop_p is None and
                (self.shape is None or self.shape == (n, h, w, groups)))

    def receive(self, p: int, douts):
        """Called from the autograd node of pass p: the gradients w.r.t. its outputs (a node may be visited more than once when its
        outputs are consumed in different sweeps)."""
        cur = self.douts[p]
        for k, d in enumerate(douts):
            if d is None:
                continue
            d = d.float().contiguous(memory_format=torch.channels_last)
            cur[k] = d if cur[k] is None else cur[k] + d
        ev = torch.cuda.Event()
        ev.record()
        self.events[p].append(ev)


class PlanBuilder:
    def __init__(self, net: "CtlNet"):
        self.net = net
        # BatchNorm groups: independent passes batched along n (each keeps its own statistics), see ctl_conv.groups
        self.groups = int(getattr(net, "_cur_groups", 1))
        self.ops: List[np.ndarray] = []
        self.b16 = bool(getattr(net, "bf16", False))
        self.pp = tuple(getattr(net, "_cur_pp", (0, 1)))                 # forward plans: pass slot (p, P) of a stacked arena, see Arena
        self.affine_mask = int(getattr(net, "_cur_affine_mask", 0))      # backward plans: the groups that add to gamma / beta gradients (0 = all)
        self.split_tail = bool(getattr(net, "_cur_split", False))        # backward plans: also cut into data-gradient chain | weight-gradient tail
        self.act = Arena(S_ACT, self.b16, self.pp)
        self.bscr = Arena(S_BSCR, self.b16)
        self.scr_bytes = 0
        self.reduce_recs: List[list] = []     # batched wgrad reduction records (one launch at the end of the plan)
        self.pending_wgrads: List[dict] = []  # deferred weight gradients (GROUP_WGRAD), emitted by flush_wgrad_groups
        self.table: Optional[np.ndarray] = None
        self.bn_log: List[tuple] = []         # (BNInfo, coefficient refs) of every training-mode BatchNorm of a forward plan, in order

    # -- low level
    def op(self, kind: int) -> np.ndarray:
        r = np.zeros((), dtype=OP_DTYPE)
        r["kind"] = kind
        r["slot"][:] = -1
        self.ops.append(r)
        return r

    @staticmethod
    def set_t(op, idx, ref):
        if ref is not None:
            op["slot"][idx] = ref[0]
            op["off"][idx] = ref[1]

    def scr(self, *nbytes):
        """Transient scratch (valid until the next op that asks for scratch)."""
        refs, off = [], 0
        for nb in nbytes:
            refs.append((S_SCR, off))
            off += _rup(int(nb), 256)
        self.scr_bytes = max(self.scr_bytes, off)
        return refs

    @staticmethod
    def P(off_floats):
        return (S_P, 4 * off_floats)

    @staticmethod
    def G(off_floats):
        return (S_GRAD, 4 * off_floats)

    # -- conv family
    def _conv_desc(self, x: T, cout, ks, stride, in_mode, hout, wout, pro, flags, act, slope, nsub=1, pad=None, dt=0):
        d = _ffi.conv_desc(n=x.n, hin=x.h, win=x.w, cin=x.c, hout=hout, wout=wout, cout=cout, ks=ks, stride=stride,
                           pad=(1 if ks in (3, 4) else 0) if pad is None else pad, in_mode=in_mode, pro_affine=1 if pro else 0,
                           pro_slope=pro[2] if pro else 0.0, epi_flags=flags, epi_act=act, epi_slope=slope, dt=dt)
        if nsub == 4:
            d["out_h"], d["out_w"], d["out_sy"], d["out_sx"], d["nsub"], d["out_sub"] = 2 * hout, 2 * wout, 2, 2, 4, 1
        d["groups"] = self.groups
        return d

    def conv(self, x: T, wp_ref, cout, ks, *, stride=1, in_mode=0, pro=None, bias_ref=None, stats=False, act=0,
             slope=0.0, res=None, accum=False, out: Optional[T] = None, arena: Optional[Arena] = None, nsub=1,
             hout=None, wout=None, bnbwd=None, pad=None, tail=None, x2=None, keep_stats=False, xout: Optional[T] = None):
        # (tail may carry a 4th element: True = the consumer wants the 2x2 sum-pool of g as well, see below)
        """Emit one CTL_OP_CONV.  pro = (scale_ref, shift_ref, slope); res = (T v, scale_ref, shift_ref).
        bnbwd = (T u, scale_ref, shift_ref, slope): data-gradient conv whose result is dL/d leaky(BN(u)); the epilogue writes
        g = result * leaky'(BN(u)) and the BatchNorm-backward sums go to the statistics partials (CTL_EPI_BNBWD).
        tail = (T block_out, T v, slope): this conv writes dL/dOut of a residual block; the epilogue stores g = dOut * leaky'(block_out)
        instead and takes the BatchNorm-backward sums of the tail (sum g, sum g*v) into the statistics partials (CTL_EPI_TAILBWD).
        x2 = (T u, coef_ref): the input is the virtual BatchNorm-backward result A*x + B*u + C (pro_affine 2, bf16 family).
        Returns (T y, stats_ref or None, stats_blocks)."""
        res2, want_pool, pool_t = None, False, None
        if tail is not None:
            assert res is None and bnbwd is None and bias_ref is None and act == 0
            stats, res2, slope = True, tail[1], tail[2]
            want_pool = len(tail) > 3 and bool(tail[3])
        if bnbwd is not None:
            assert res is None and not accum and bias_ref is None and act == 0
            stats, res, slope = True, (bnbwd[0], bnbwd[1], bnbwd[2]), bnbwd[3]
        if hout is None:
            if in_mode == _ffi.IN_UP2:
                hout, wout = 2 * x.h, 2 * x.w
            elif stride == 2:
                hout, wout = ((x.h + 1) // 2, (x.w + 1) // 2) if ks == 3 else (x.h // 2, x.w // 2)
            else:
                hout, wout = x.h, x.w
        flags = (_ffi.EPI_BIAS if bias_ref is not None else 0) | (_ffi.EPI_STATS if stats else 0) | \
                ((_ffi.EPI_BNBWD if bnbwd is not None else _ffi.EPI_RES) if res is not None else 0) | (_ffi.EPI_ACCUM if accum else 0) | \
                (_ffi.EPI_TAILBWD if tail is not None else 0)
        oh, ow = (2 * hout, 2 * wout) if nsub == 4 else (hout, wout)
        if out is None:
            out = (arena or self.act).tensor(x.n, oh, ow, cout)
        assert (out.n, out.h, out.w, out.c) == (x.n, oh, ow, cout), (out, x.n, oh, ow, cout)
        dt = _ffi.DT_X3 if (wp_ref[0] == S_WP and wp_ref[1] // 4 in self.net._x3_packs) else 0      # (the pack's layout decides: see CtlNet._finalize_storage)
        if self.b16:        # bf16 MFMA family; which of x / y / res is STORED as bf16 follows from where the tensor lives
            assert bnbwd is None or (x.b16 and out.b16 and bnbwd[0].b16), "CTL_EPI_BNBWD (bf16): x, y and u must be bf16-stored"
            assert tail is None or (out.b16 and tail[0].b16 and tail[1].b16), "CTL_EPI_TAILBWD (bf16): y, out and v must be bf16-stored"
            dt = _ffi.DT_BF16 | (_ffi.DT_X16 if x.b16 else 0) | (_ffi.DT_Y16 if out.b16 else 0) | \
                (_ffi.DT_RES16 if (res is not None and res[0].b16) or tail is not None else 0)
        d = self._conv_desc(x, cout, ks, stride, in_mode, hout, wout, pro, flags, act, slope, nsub, pad, dt)
        if x2 is not None:
            assert pro is None and x.b16 == x2[0].b16 == self.b16 and (x2[0].n, x2[0].h, x2[0].w, x2[0].c) == (x.n, x.h, x.w, x.c)
            d["pro_affine"] = 2
            pro = (x2[1], None)
        op = self.op(_ffi.OP_CONV)
        op["i"][:CONV_WORDS] = np.frombuffer(d.tobytes(), dtype="<i4")
        if want_pool and FUSE_POOL and lib.ctl_conv_pool_ok(_ffi.desc_ptr(d)):
            # the tail epilogue also writes sumpool2(g) for the consuming nearest-upsample block (its 1x1 weight / data gradients)
            pool_t = self.bscr.tensor(out.n, out.h // 2, out.w // 2, out.c)
        self.last_pool = pool_t
        stats_ref, blocks = None, 0
        if stats:
            blocks = lib.ctl_conv_stats_blocks(_ffi.desc_ptr(d))
            if blocks <= 0:
                raise _ffi.CtlError("conv plan: " + lib.ctl_last_error().decode())
            if tail is not None or keep_stats:      # consumed by the NEXT block's finalize: kept in the backward arena, not in the transient scratch
                stats_ref = self.bscr.alloc(4 * self.groups * blocks * 2 * cout)
            else:
                stats_ref, = self.scr(4 * self.groups * blocks * 2 * cout)
        for idx, ref in enumerate([x.ref, wp_ref, bias_ref, pro[0] if pro else None, pro[1] if pro else None,
                                   (tail[0].ref if tail is not None else (res[0].ref if res else None)), res[1] if res else None, res[2] if res else None,
                                   out.ref, stats_ref, res2.ref if res2 is not None else None, x2[0].ref if x2 is not None else None,
                                   pool_t.ref if pool_t is not None else None, xout.ref if xout is not None else None]):
            self.set_t(op, idx, ref)
        assert xout is None or (x2 is not None and (xout.n, xout.h, xout.w, xout.c, xout.b16) == (x.n, x.h, x.w, x.c, x.b16))
        return out, stats_ref, blocks

    def wgrad(self, x: T, dy: T, ks, *, stride=1, in_mode=0, pro=None, dw_ref, strides, dbias_ref=None, accumulate=False, dy2=None):
        """CTL_OP_WGRAD + CTL_OP_WGRAD_REDUCE for the conv x -> dy.  dy2 = (T u, coef_ref): the output gradient is the virtual
        BatchNorm-backward result A*dy + B*u + C (bf16 family)."""
        assert dy2 is None or (dy.b16 == dy2[0].b16 == self.b16 and ks == 3 and stride == 1)
        dt = (_ffi.DT_BF16 | (_ffi.DT_X16 if x.b16 else 0) | (_ffi.DT_Y16 if dy.b16 else 0)) if self.b16 else 0
        if (X3 and X3_WGRAD and not self.b16 and x.c % 16 == 0 and dy.c % 16 == 0 and
                ((ks == 3 and in_mode in (_ffi.IN_PLAIN, _ffi.IN_UP2) and (stride == 1 or in_mode == _ffi.IN_PLAIN)) or (ks == 2 and stride == 2 and in_mode == _ffi.IN_PLAIN))):
            dt = _ffi.DT_X3          # the weight gradient on the bf16 matrix pipe over the exact three-way split (csrc/ctl_wgrad_x3.hip)
        d = self._conv_desc(x, dy.c, ks, stride, in_mode, dy.h, dy.w, pro, 0, 0, 0.0, dt=dt)
        dp = _ffi.desc_ptr(d)
        assert dw_ref[0] == S_GRAD and (dbias_ref is None or dbias_ref[0] == S_GRAD)
        refs = [x.ref, pro[0] if pro else None, pro[1] if pro else None, dy.ref, None, None, dy2[0].ref if dy2 else None, dy2[1] if dy2 else None]
        if GROUP_WGRAD and (dt == _ffi.DT_X3 or (GROUP_WGRAD_BF16 and self.b16)):
            cls = int(lib.ctl_wgrad_group_class(dp, 1 if dy2 else 0))
            if cls >= 0:
                self.pending_wgrads.append(dict(cls=cls, d=d, refs=refs, dw_ref=dw_ref, dbias_ref=dbias_ref, strides=[int(v) for v in strides],
                                                accumulate=bool(accumulate), cin=x.c, cout=dy.c, ks=ks))
                return
        self._emit_wgrad(d, refs, dw_ref, dbias_ref, strides, accumulate, x.c, dy.c, ks, None)

    def _emit_wgrad(self, d, refs, dw_ref, dbias_ref, strides, accumulate, cin, cout, ks, splits):
        """One WGRAD record + its reduction record.  splits = None: a launch of its own (the library's split count); else a member of a
        grouped launch with that many pixel splits (record word i[24])."""
        dp = _ffi.desc_ptr(d)
        cin_p, cout_p = _rup(cin, 16), _rup(cout, 16)
        if splits is None:
            wb, bb = 4 * lib.ctl_wgrad_partial_floats(dp), 4 * lib.ctl_wgrad_bias_partial_floats(dp)
        else:
            wb, bb = 4 * splits * ks * ks * cin_p * cout_p, 4 * splits * cout_p
        if wb == 0:
            raise _ffi.CtlError("wgrad plan: " + lib.ctl_last_error().decode())
        # every layer keeps its own partial buffers (backward arena): all reductions run as ONE table-driven launch at the end
        wref = self.bscr.alloc(wb)
        bref = self.bscr.alloc(bb) if dbias_ref is not None else None
        words = np.frombuffer(d.tobytes(), dtype="<i4")
        op = self.op(_ffi.OP_WGRAD)
        op["i"][:CONV_WORDS] = words
        refs = list(refs)
        refs[4], refs[5] = wref, bref
        for idx, ref in enumerate(refs):
            self.set_t(op, idx, ref)
        if splits is None:
            splits = lib.ctl_wgrad_splits(dp)
        else:
            op["i"][CONV_WORDS] = splits
        self.reduce_recs.append([wref[1] // 4, bref[1] // 4 if bref else -1, dw_ref[1] // 4, dbias_ref[1] // 4 if dbias_ref else -1,
                                 splits, (ks * ks) | (ks << 8), cin, cout, cin_p, cout_p, *[int(v) for v in strides],
                                 1 if accumulate else 0, 0])

    def flush_wgrad_groups(self):
        """Emit the deferred weight gradients: per class (kernel instantiation) in groups of up to WGRAD_GROUP_MAX members, each group
        one CTL_OP_WGRAD_GROUP record followed by its members' WGRAD records; a class with a single member is launched on its own."""
        pend, self.pending_wgrads = self.pending_wgrads, []
        by_cls: Dict[int, list] = {}
        for w in pend:
            by_cls.setdefault(w["cls"], []).append(w)
        for cls in sorted(by_cls):
            members = by_cls[cls]
            for g0 in range(0, len(members), _ffi.WGRAD_GROUP_MAX):
                grp = members[g0:g0 + _ffi.WGRAD_GROUP_MAX]
                if len(grp) == 1:
                    w = grp[0]
                    self._emit_wgrad(w["d"], w["refs"], w["dw_ref"], w["dbias_ref"], w["strides"], w["accumulate"], w["cin"], w["cout"], w["ks"], None)
                    continue
                descs = np.concatenate([np.atleast_1d(w["d"]) for w in grp])
                splits = np.zeros(len(grp), dtype=np.int32)
                _ffi.check(lib.ctl_wgrad_group_plan(descs.ctypes.data, len(grp), splits.ctypes.data), "ctl_wgrad_group_plan")
                op = self.op(_ffi.OP_WGRAD_GROUP)
                op["i"][0] = len(grp)
                for w, sp in zip(grp, splits):
                    self._emit_wgrad(w["d"], w["refs"], w["dw_ref"], w["dbias_ref"], w["strides"], w["accumulate"], w["cin"], w["cout"], w["ks"], int(sp))

    def flush_wgrad_reductions(self):
        self.flush_wgrad_groups()
        if not self.reduce_recs:
            return
        assert self.table is None
        self.table = np.asarray(self.reduce_recs, dtype=np.int64)
        op = self.op(_ffi.OP_WGRAD_REDUCE_BATCH)
        op["i"][0] = len(self.reduce_recs)
        op["l"][0] = max(-(-((r[5] & 0xff) * r[6] * r[7] + r[7]) // (64 if r[4] <= 64 else 8)) for r in self.reduce_recs)
        self.set_t(op, 0, (S_BSCR, 0))
        self.set_t(op, 1, (S_GRAD, 0))
        self.set_t(op, 2, (S_TAB, 0))
        self.reduce_recs = []

    # -- BatchNorm
    def bn_forward(self, bn: BNInfo, stats_ref, blocks, count, mode: str):
        """Returns dict(scale, shift, mean, invstd) of refs living in the ACT arena."""
        c, G = bn.c, self.groups
        assert count % G == 0
        count //= G                                   # the statistics of one group
        co = {k: self.act.rows(c, G) for k in ("scale", "shift", "mean", "invstd", "uvar")}      # [groups][c] each
        if mode == "C":
            op = self.op(_ffi.OP_BN_EVAL)
            op["i"][0], op["i"][1] = c, G
            op["f"][0] = EPS
            for idx, ref in enumerate([self.P(bn.g_off), self.P(bn.b_off), (S_B, 4 * bn.rm_off), (S_B, 4 * bn.rv_off),
                                       co["scale"], co["shift"]]):
                self.set_t(op, idx, ref)
            return co
        op = self.op(_ffi.OP_BN_FINALIZE)
        op["i"][0], op["i"][1], op["i"][2], op["i"][3] = blocks, c, 1 if mode == "A" else 0, G
        op["l"][0] = count
        op["f"][0], op["f"][1] = EPS, MOMENTUM
        for idx, ref in enumerate([stats_ref, self.P(bn.g_off), self.P(bn.b_off), (S_B, 4 * bn.rm_off),
                                   (S_B, 4 * bn.rv_off), (S_NBT, 8 * bn.nbt_idx), co["scale"], co["shift"], co["mean"],
                                   co["invstd"], co["uvar"]]):
            self.set_t(op, idx, ref)
        self.bn_log.append((bn, co))
        return co

    @staticmethod
    def mask(*ts) -> int:
        """bf16 storage mask over an element-wise op's tensor arguments (ctl_*_dt)."""
        return sum(1 << k for k, t in enumerate(ts) if t is not None and t.b16)

    def bn_act(self, x: T, co, slope, out: T):
        op = self.op(_ffi.OP_BN_ACT)
        op["i"][0], op["i"][1] = x.c, self.groups
        op["i"][25] = self.mask(x, out)
        op["l"][0] = x.n * x.h * x.w
        op["f"][0] = slope
        for idx, ref in enumerate([x.ref, co["scale"], co["shift"], out.ref]):
            self.set_t(op, idx, ref)

    def bn_backward(self, mode_kind: int, dy: T, act_src: Optional[T], bn_src: T, bn: BNInfo, co, slope, *, ds: Optional[T],
                    dx: T, affine_grad: bool):
        """reduce -> finalize -> apply.  mode_kind 0 = residual tail, 1 = BN->activation tail.
        dx = None (mode 0 with ds): no apply pass -- the consumers take (ds, bn_src, coef) through their BatchNorm-backward prologue
        (conv(x2=...), wgrad(dy2=...)).  Returns the coefficient ref."""
        c, pixels, G = bn_src.c, bn_src.n * bn_src.h * bn_src.w, self.groups
        assert dx is not None or (mode_kind == 0 and ds is not None)
        part, = self.scr(4 * G * _ffi.RED_BLOCKS * 2 * c)
        coef = self.bscr.alloc(4 * 3 * c * G)
        # residual tail (mode 0), fp32: the reduction also writes ds = dy * leaky'(act_src) (it has it in registers), and the apply pass
        # then runs in mode 2 on ds -- same arithmetic bit for bit, one tensor read less.  (bf16 keeps the two-operand apply: reading a
        # ROUNDED ds back would add a rounding point.)
        ds_early = mode_kind == 0 and ds is not None and (not self.b16 or dx is None)
        op = self.op(_ffi.OP_BWD_REDUCE)
        op["i"][0], op["i"][1], op["i"][2] = mode_kind, c, G
        op["i"][25] = self.mask(dy, act_src, bn_src, ds if ds_early else None)
        op["l"][0] = pixels
        op["f"][0] = slope
        for idx, ref in enumerate([dy.ref, act_src.ref if act_src else None, bn_src.ref, co["scale"], co["shift"], part,
                                   ds.ref if ds_early else None]):
            self.set_t(op, idx, ref)
        op = self.op(_ffi.OP_BN_BWD_FINALIZE)
        op["i"][0], op["i"][1], op["i"][2], op["i"][4] = c, 0, G, self.affine_mask
        op["l"][0] = pixels // G
        for idx, ref in enumerate([part, self.P(bn.g_off), co["mean"], co["invstd"], coef,
                                   self.G(bn.g_off) if affine_grad else None, self.G(bn.b_off) if affine_grad else None]):
            self.set_t(op, idx, ref)
        if dx is None:
            return coef
        op = self.op(_ffi.OP_BWD_APPLY)
        if ds_early:
            op["i"][0], op["i"][1], op["i"][2] = 2, c, G
            op["i"][25] = self.mask(ds, None, bn_src, None, dx)
            op["l"][0] = pixels
            for idx, ref in enumerate([ds.ref, None, bn_src.ref, None, None, coef, None, dx.ref]):
                self.set_t(op, idx, ref)
            return coef
        op["i"][0], op["i"][1], op["i"][2] = mode_kind, c, G
        op["i"][25] = self.mask(dy, act_src, bn_src, ds, dx)
        op["l"][0] = pixels
        op["f"][0] = slope
        for idx, ref in enumerate([dy.ref, act_src.ref if act_src else None, bn_src.ref, co["scale"], co["shift"], coef,
                                   ds.ref if ds else None, dx.ref]):
            self.set_t(op, idx, ref)
        return coef

    def bn_backward_from_stats(self, g: T, bn_src: T, bn: BNInfo, co, stats_ref, blocks, *, dx: T, affine_grad: bool):
        """BatchNorm backward whose reduction already happened in the producing conv (conv(..., bnbwd=...)): finalize + apply.
        dx = None: finalize only (the consumers apply the coefficients in their staging).  Returns the coefficient ref."""
        c, pixels, G = bn_src.c, bn_src.n * bn_src.h * bn_src.w, self.groups
        coef = self.bscr.alloc(4 * 3 * c * G)
        op = self.op(_ffi.OP_BN_BWD_FINALIZE)
        op["i"][0], op["i"][1], op["i"][2], op["i"][3], op["i"][4] = c, 0, G, blocks, self.affine_mask
        op["l"][0] = pixels // G
        for idx, ref in enumerate([stats_ref, self.P(bn.g_off), co["mean"], co["invstd"], coef,
                                   self.G(bn.g_off) if affine_grad else None, self.G(bn.b_off) if affine_grad else None]):
            self.set_t(op, idx, ref)
        if dx is None:
            return coef
        op = self.op(_ffi.OP_BWD_APPLY)
        op["i"][0], op["i"][1], op["i"][2] = 2, c, G
        op["i"][25] = self.mask(g, None, bn_src, None, dx)
        op["l"][0] = pixels
        for idx, ref in enumerate([g.ref, None, bn_src.ref, None, None, coef, None, dx.ref]):
            self.set_t(op, idx, ref)
        return coef

    def chan_sum(self, dy: T, out_ref):
        c = dy.c
        part, = self.scr(4 * _ffi.RED_BLOCKS * 2 * c)
        op = self.op(_ffi.OP_BWD_REDUCE)
        op["i"][0], op["i"][1] = 2, c
        op["i"][25] = self.mask(dy)
        op["l"][0] = dy.n * dy.h * dy.w
        self.set_t(op, 0, dy.ref)
        self.set_t(op, 5, part)
        op = self.op(_ffi.OP_CHAN_SUM_FINALIZE)
        op["i"][0], op["i"][1] = c, 0
        self.set_t(op, 0, part)
        self.set_t(op, 1, out_ref)

    def sumpool2(self, dup: T, dx: T, accumulate=False):
        op = self.op(_ffi.OP_SUMPOOL2)
        op["i"][:5] = [dx.n, dx.h, dx.w, dx.c, 1 if accumulate else 0]
        op["i"][25] = self.mask(dup, dx)
        self.set_t(op, 0, dup.ref)
        self.set_t(op, 1, dx.ref)

    def sigmoid_bwd(self, dy: T, y: T, dx: T):
        assert not (dy.b16 or y.b16 or dx.b16), "sigmoid_bwd works on the fp32 network output"
        op = self.op(_ffi.OP_SIGMOID_BWD)
        op["l"][0] = y.n * y.h * y.w * y.c
        for idx, ref in enumerate([dy.ref, y.ref, dx.ref]):
            self.set_t(op, idx, ref)

    def dropout(self, z: T, out: T, p: float, *, keep_in=None, keep_out=None, salt: int = 0):
        """nn.Dropout2d (whole channels per sample, scaled by 1/(1-p)).  keep_in = ref of a given [n][c] pattern (backward pass, injected
        patterns); otherwise the pattern is drawn on the device from the network's RNG state (S_STATE) and this call site's salt and
        written to keep_out."""
        op = self.op(_ffi.OP_DROPOUT2D)
        op["i"][25] = self.mask(z, out)
        op["i"][0], op["i"][1], op["i"][2] = z.n, z.h * z.w, z.c
        op["f"][0] = p
        op["l"][0] = salt
        for idx, ref in enumerate([z.ref, keep_in, None if keep_in is not None else (S_STATE, 0), out.ref, keep_out]):
            self.set_t(op, idx, ref)

    def zero(self, ref, nbytes):
        op = self.op(_ffi.OP_ZERO)
        op["l"][0] = nbytes
        self.set_t(op, 0, ref)

    def copy(self, src_ref, dst_ref, nbytes):
        op = self.op(_ffi.OP_COPY)
        op["l"][0] = nbytes
        self.set_t(op, 0, src_ref)
        self.set_t(op, 1, dst_ref)

    def finish(self, rec=None, out_shapes=None) -> Plan:
        self.flush_wgrad_reductions()
        p = Plan()
        p.replay = None
        p.table_np, p.table_dev = self.table, None
        p.ops = np.stack(self.ops) if self.ops else np.zeros(0, dtype=OP_DTYPE)
        p.ops = np.ascontiguousarray(p.ops)
        p.n_ops = len(self.ops)
        p.main_ops = p.tail_ops = None
        if self.split_tail:
            # The weight-gradient family (and the channel sums of the transposed convs' bias gradients, reduction + finalize) in plan order
            # behind everything else in plan order.  Safe for the reasons the grouped weight gradients are deferred (GROUP_WGRAD): nothing in
            # the plan reads dW, arenas are bump-allocated, and every in-place pass on a gradient tensor precedes the weight gradient reading it
            tail = []
            for k, o in enumerate(self.ops):
                kind = int(o["kind"])
                if kind in _TAIL_KINDS:
                    tail.append(k)
                    if kind == _ffi.OP_CHAN_SUM_FINALIZE:      # its reduction (BWD_REDUCE mode 2 into the transient scratch) goes with it
                        assert int(self.ops[k - 1]["kind"]) == _ffi.OP_BWD_REDUCE and int(self.ops[k - 1]["i"][0]) == 2
                        tail.insert(len(tail) - 1, k - 1)
            ts = set(tail)
            main = [k for k in range(len(self.ops)) if k not in ts]
            p.main_ops = np.ascontiguousarray(p.ops[main])
            p.tail_ops = np.ascontiguousarray(p.ops[tail]) if tail else None
        p.act_bytes, p.scr_bytes, p.bscr_bytes = self.act.size, self.scr_bytes, self.bscr.size
        p.rec, p.out_shapes = rec, out_shapes
        p.groups = self.groups
        p.bn_log = self.bn_log
        return p


# ================================================================================================ network base class
class CtlNet(nn.Module):
    """Flat-storage network whose compute is a compiled plan of HIP kernel launches."""

    def __init__(self, spec, cin: int, device, bf16: bool = False):
        super().__init__()
        self._spec = spec
        self.cin = cin
        # BASELINE config 3: network-internal activations and gradients stored as bf16, convolutions on v_mfma_f32_16x16x32_bf16 with
        # fp32 accumulation; master weights, BatchNorm statistics / coefficients, parameter gradients and network inputs / outputs fp32
        self.bf16 = bool(bf16)
        self._bn_track = True              # False inside `disable_tracking_bn_stats` (mode B)
        self._convs: Dict[str, ConvInfo] = {}
        self._bns: Dict[str, BNInfo] = {}
        self._plans: Dict[tuple, Plan] = {}
        self._packed_ok = False
        self._grad_written = True          # see zero_grad / FlatAdam.step
        self._grad_is_zero = False         # the gradient buffer is KNOWN to hold zeros (zero_grad ran, nothing wrote since): see zero_grad
        # deferred parameter gradients (solver.cooperative_step): while set, a backward pass parks its flat gradient here instead of
        # handing it to autograd; collect_deferred_grads() adds them into `.grad` with ONE launch, in forward order
        self._defer_grads = False
        self._deferred: list = []
        self._pass_seq = 0
        # data parallelism (dist.py): backward passes of this step that still owe a parameter gradient, and what to call when the last
        # one has been issued (the network's range of the gradient bucket is complete: its all-reduce can start)
        self._pending_bwd = 0
        self._on_grads_complete = None
        self._scr: Optional[dict] = None          # stream handle -> scratch tensor
        self._arenas = ArenaPool()
        self._stack: Optional[PassStack] = None   # set by the solver for the duration of one step: passes whose backward runs stacked
        # nn.Dropout2d behind every residual block (encoder_decoder.py:58-66; `encoder_dropout` / `decoder_dropout`, None upstream's default)
        self.drop_p: Optional[float] = None
        self._drop_state: Optional[torch.Tensor] = None      # device int64[3]: seed, pass counter (advanced by ctl_step_tick), unused
        self._drop_keep: Optional[torch.Tensor] = None       # injected patterns (tests): the blocks' [n][c] rows concatenated
        self._build_tree()
        self._finalize_storage(torch.device(device))

    # ---------------------------------------------------------------- construction / storage
    def _build_tree(self):
        for key, kind, a in self._spec:
            parts = key.split(".")
            m = self
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, _Holder())
                m = m._modules[p]
            if kind == "bn":
                leaf = _BNP(a[0])
            else:
                leaf = _ConvP(a[0], a[1], a[2], transposed=(kind == "convT"))
            m.add_module(parts[-1], leaf)

    def _finalize_storage(self, device: torch.device):
        old_p = {n: p.detach().to("cpu") for n, p in self.named_parameters()}
        old_b = {n: b.detach().to("cpu") for n, b in self.named_buffers()}
        poff, off = {}, 0
        for n, p in self.named_parameters():
            poff[n] = off
            off += _rup(p.numel(), ALIGN_F)
        self._poff, self._pcount = poff, off
        flat = torch.zeros(off, dtype=torch.float32, device=device)
        grad = torch.zeros(off, dtype=torch.float32, device=device)
        for n, p in self.named_parameters():
            v = flat[poff[n]:poff[n] + p.numel()].view(p.shape)
            v.copy_(old_p[n])
            p.data = v
            p.grad = grad[poff[n]:poff[n] + p.numel()].view(p.shape)
        self._flat_data = flat
        self._flat = flat.detach().requires_grad_(True)      # the autograd leaf (shares storage)
        self._flat.grad = grad
        boff, off, nbt_names = {}, 0, []
        for n, b in self.named_buffers():
            if b.dtype == torch.long:
                nbt_names.append(n)
            else:
                boff[n] = off
                off += _rup(b.numel(), ALIGN_F)
        self._bflat = torch.zeros(max(off, 1), dtype=torch.float32, device=device)
        self._nbt = torch.zeros(max(len(nbt_names), 1), dtype=torch.long, device=device)
        mods = dict(self.named_modules())
        for n in list(boff) + nbt_names:
            mname, bname = n.rsplit(".", 1)
            if n in boff:
                v = self._bflat[boff[n]:boff[n] + old_b[n].numel()].view(old_b[n].shape)
            else:
                v = self._nbt[nbt_names.index(n)]
            v.copy_(old_b[n])
            mods[mname]._buffers[bname] = v
        self._boff, self._nbt_names = boff, nbt_names
        # op-level descriptors + packed-weight layout
        wp = 0
        self._x3_packs = set()             # float offsets (into the packed-weight buffer) of the packs in the three-plane X3 layout
        x3_on = X3 and not self.bf16

        def wp_floats(cin_eff, cout_eff, ks):
            """(float count of one packed sub-problem, X3 layout?) of an effective conv cin_eff -> cout_eff"""
            if x3_on and ks >= 2 and cin_eff % 16 == 0 and (cout_eff % 16 == 0 or cout_eff in (4, 8, 12)):
                return lib.ctl_conv_wpack_floats_x3(cin_eff, cout_eff, ks), True
            return lib.ctl_conv_wpack_floats(cin_eff, cout_eff, ks), False

        def take(cin_eff, cout_eff, ks, count=1):
            nonlocal wp
            n, x3 = wp_floats(cin_eff, cout_eff, ks)
            off, wp = wp, wp + count * n
            if x3:
                self._x3_packs.update(off + z * n for z in range(count))
            return off, n
        for key, kind, a in self._spec:
            if kind == "bn":
                bi = BNInfo()
                bi.key, bi.c = key, a[0]
                bi.g_off, bi.b_off = poff[key + ".weight"], poff[key + ".bias"]
                bi.rm_off, bi.rv_off = boff[key + ".running_mean"], boff[key + ".running_var"]
                bi.nbt_idx = nbt_names.index(key + ".num_batches_tracked")
                self._bns[key] = bi
            else:
                ci = ConvInfo()
                ci.key, ci.cin, ci.cout, ci.ks, ci.transposed = key, a[0], a[1], a[2], kind == "convT"
                ci.w_off, ci.b_off = poff[key + ".weight"], poff[key + ".bias"]
                if ci.transposed:     # forward = 4 scattered 1x1 problems, dgrad = 2x2 stride-2 conv
                    ci.wp_fwd, ci.wp_sub = take(ci.cin, ci.cout, 1, 4)
                    ci.wp_dgrad, _ = take(ci.cout, ci.cin, 2)
                else:
                    ci.wp_sub = 0
                    ci.wp_fwd, _ = take(ci.cin, ci.cout, ci.ks)
                    ci.wp_dgrad, _ = take(ci.cout, ci.cin, ci.ks)
                ci.wp_up = -1
                if getattr(self, "up_type", None) == "NN" and ci.ks == 3 and key.startswith("up") and key.endswith(".conv.0"):
                    # first conv of a nearest-upsample block: its data gradient followed by the upsample backward (2x2 sum-pool)
                    # is ONE 4x4 stride-2 conv over dU (16 taps per low-res pixel instead of 36 + a full-resolution round trip)
                    ci.wp_up, _ = take(ci.cout, ci.cin, 4)
                ci.wp_upf = ci.wp_s2d = ci.wp_c4 = -1
                ci.wp_ph = 0
                if SMALL_CIN and not self.bf16 and ci.ks == 3 and not ci.transposed and ci.cin <= 4:
                    # first layer (image: 1 channel, STN input: 4): the 3x3 taps are K-packed, 3 fragments per cout tile
                    ci.wp_c4, wp = wp, wp + ((ci.cout + 15) // 16) * 3 * 256
                if ci.wp_up >= 0 and PHASE_CONVS:
                    # ... and its forward on the nearest-upsampled input is four 2x2 phase convs on the stored input
                    ci.wp_upf, ci.wp_ph = take(ci.cin, ci.cout, 2, 4)
                if PHASE_CONVS and ci.ks == 3 and not ci.transposed and key.endswith(".down"):
                    # stride-2 conv of a down block: its data gradient is four phase convs with <= 2x2 taps over dy (instead of a
                    # 3x3 conv over a zero-inserted tensor that is 75 % zeros)
                    ci.wp_s2d, ci.wp_ph = take(ci.cout, ci.cin, 2, 4)
                self._convs[key] = ci
        self._wp = torch.zeros(max(wp, 1), dtype=torch.float32, device=device)
        self._pack_plan = self._build_pack_plan()
        self._packed_ok = False
        self._plans.clear()
        self._scr = None

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self._flat_data.device))
        if probe.dtype != torch.float32:
            raise _ffi.CtlError("CtlNet holds fp32 master weights; dtype conversion is not supported")
        if probe.device != self._flat_data.device:
            self._finalize_storage(probe.device)
        return self

    @property
    def device(self):
        return self._flat_data.device

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._packed_ok = False
        return r

    def bind_grad_buffer(self, buf: torch.Tensor):
        """Re-home this network's flat gradient inside a caller-owned bucket (one all-reduce for all five networks)."""
        assert buf.numel() == self._pcount and buf.dtype == torch.float32 and buf.device == self._flat_data.device
        buf.copy_(self._flat.grad)
        self._flat.grad = buf
        self._grad_written = True
        self._grad_is_zero = False
        for n, p in self.named_parameters():
            p.grad = buf[self._poff[n]:self._poff[n] + p.numel()].view(p.shape)

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are views of one flat buffer that is never re-allocated: zero it in place.  (upstream calls
        `decoder_function.zero_grad()` inside the masking functions, model_util.py:251-254)"""
        # A buffer nothing has written since the last zero_grad is not filled again (upstream zeroes the decoders after every saliency
        # pass, the step zeroes them once more: 17 of 22 fills per targeted step found nothing to clear).  Every writer inside the
        # engine clears the flag; anything else that writes `.grad` says so through mark_grad_written().
        if not self._grad_is_zero:
            self._flat.grad.zero_()
            self._grad_is_zero = True
        self._grad_written = False        # FlatAdam skips a network no backward pass has written since (torch: `.grad is None`)

    def collect_deferred_grads(self):
        """Add the parked per-pass gradients into the gradient buffer (fixed order: the order of the forward passes) on the current
        stream, which first waits for the streams that produced them."""
        if not self._deferred:
            return
        cur = torch.cuda.current_stream()
        items = sorted(self._deferred, key=lambda it: it[0])
        self._deferred = []
        for _, g, ev in items:
            cur.wait_event(ev)
        for i in range(0, len(items), 8):
            chunk = [g for _, g, _ in items[i:i + 8]]
            arr = (ctypes.c_void_p * len(chunk))(*[g.data_ptr() for g in chunk])
            check(lib.ctl_accumulate(self._flat.grad.data_ptr(), arr, len(chunk), self._pcount, cur.cuda_stream), "ctl_accumulate")
            for g in chunk:
                g.record_stream(cur)
        self._grad_written = True
        self._grad_is_zero = False

    def mark_grad_written(self):
        """Call after filling `.grad` by other means than a backward pass of this network (e.g. a hand-written gradient)."""
        self._grad_written = True
        self._grad_is_zero = False

    def weights_changed(self):
        """Call after modifying parameters in place by other means than the engine's optimizer / load_state_dict."""
        self._packed_ok = False
        self._wepoch = getattr(self, "_wepoch", 0) + 1
        self._last_pass = None

    # ---------------------------------------------------------------- a forward pass whose activations can serve a second, identical pass
    def remember_pass(self, x, act, outs, plan, mode, groups):
        """Called by the autograd bridge after every forward pass.  A training-mode, tracking ('A') pass is kept: a later request to run
        the SAME input through the SAME weights in the same mode (the saliency forward of the targeted latent masks, model_util.py:214,
        decodes the code the standard pass has just decoded) re-uses its activations instead of recomputing them (`reuse_pass`)."""
        self._last_pass = ((x, x._version, act, outs, plan, getattr(self, "_wepoch", 0), tuple(o._version for o in outs))
                           if (mode == "A" and groups == 1) else None)

    def forget_pass(self):
        self._last_pass = None

    def reuse_pass(self, x: torch.Tensor):
        """(outputs, backward) of the remembered pass if running `x` through the network NOW would repeat it exactly -- same storage, same
        tensor version, same weights, training mode with tracking -- else None.  backward(douts) is the data-gradient-only backward
        (frozen weights); the caller accounts for the second running-statistics update with `replay_running_stats`."""
        lp = getattr(self, "_last_pass", None)
        if lp is None or self.bn_mode() != "A" or self.drop_p is not None:
            return None
        x0, ver, act, outs, plan, wepoch, out_vers = lp
        # (x0 is held by the record, so its storage cannot have been handed to another tensor: equal pointers mean the same tensor;
        #  an output modified in place since the pass -- the backward reads the outputs -- voids the record as well)
        if (x0.data_ptr() != x.data_ptr() or tuple(x0.shape) != tuple(x.shape) or x0.stride() != x.stride() or x0._version != ver or
                wepoch != getattr(self, "_wepoch", 0) or not self._packed_ok or tuple(o._version for o in outs) != out_vers):
            return None

        def backward(douts):
            dx, _ = self.run_backward(x0, act, outs, plan, "A", tuple(douts), need_dx=True, need_w=False, affine=True)
            return dx
        return outs, backward, (act, plan)

    def replay_running_stats(self, handle):
        """The running-statistics update of every BatchNorm of a remembered pass, once more (ctl_bn_replay_running): ONE launch."""
        act, plan = handle
        if not plan.bn_log:
            return
        if plan.replay is None:
            recs = [[co["mean"][1], co["uvar"][1], bn.rm_off, bn.rv_off, bn.nbt_idx, bn.c] for bn, co in plan.bn_log]
            pb = PlanBuilder(self)
            pb.table = np.asarray(recs, dtype=np.int64)
            op = pb.op(_ffi.OP_BN_REPLAY)
            op["i"][0] = len(recs)
            op["f"][0] = MOMENTUM
            for idx, ref in enumerate([(S_ACT, 0), (S_B, 0), (S_NBT, 0), (S_TAB, 0)]):
                pb.set_t(op, idx, ref)
            plan.replay = pb.finish()
        self._run(plan.replay, {S_ACT: act.t, S_B: self._bflat, S_NBT: self._nbt})

    def param_list(self):
        """The network's parameters as a cached list (the module tree is fixed after construction): `Module.parameters()` walks ~100
        sub-modules on every call, and the training loop asks several hundred times per step (set_grad, wants_param_grad) -- 2 ms of
        host time per step before this cache (tools/debug/host_profile.py)."""
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = self.__dict__["_plist"] = list(self.parameters())
        return pl

    def wants_param_grad(self) -> bool:
        return any(p.requires_grad for p in self.param_list())

    def train(self, mode: bool = True):
        """nn.Module.train, without the walk over the module tree when the whole tree is already in that mode (it is flipped only here)."""
        if self.__dict__.get("_mode_synced") is mode and self.training is mode:
            return self
        super().train(mode)
        self.__dict__["_mode_synced"] = mode
        return self

    # ---------------------------------------------------------------- weight packing
    def _build_pack_plan(self) -> Plan:
        pb = PlanBuilder(self)
        recs = []

        def pack(src_off_f, dst_off_f, cout, cin, ks, strides, flip, mode=0):
            if dst_off_f in self._x3_packs:        # three bf16 planes per fragment for the CTL_DT_X3 launches (ctl_pack_weights_x3_batched)
                total, mode = lib.ctl_conv_wpack_floats_x3(cin, cout, ks), mode | _ffi.PACK_X3
            else:
                total = lib.ctl_conv_wpack_floats(cin, cout, ks)
            recs.append([src_off_f, dst_off_f, cout, cin, ks, int(flip), *strides, total, mode])

        for ci in self._convs.values():
            k2 = ci.ks * ci.ks
            if ci.transposed:      # weight [Cin][Cout][2][2]
                for z in range(4):
                    pack(ci.w_off + z, ci.wp_fwd + z * ci.wp_sub, ci.cout, ci.cin, 1, (4, ci.cout * 4, 0, 0), False)
                pack(ci.w_off, ci.wp_dgrad, ci.cin, ci.cout, 2, (ci.cout * 4, 4, 2, 1), False)
            else:                  # weight [Cout][Cin][ks][ks]
                pack(ci.w_off, ci.wp_fwd, ci.cout, ci.cin, ci.ks, (ci.cin * k2, k2, ci.ks, 1), False)
                pack(ci.w_off, ci.wp_dgrad, ci.cin, ci.cout, ci.ks, (k2, ci.cin * k2, ci.ks, 1), True)
                if ci.wp_up >= 0:      # mode 1: the 4x4 kernel is summed from the 3x3 taps inside the pack kernel
                    pack(ci.w_off, ci.wp_up, ci.cin, ci.cout, 4, (k2, ci.cin * k2, ci.ks, 1), False, mode=1)
                if ci.wp_c4 >= 0:      # mode 4: K-packed first-layer weights
                    recs.append([ci.w_off, ci.wp_c4, ci.cout, ci.cin, 3, 0, ci.cin * k2, k2, ci.ks, 1, ((ci.cout + 15) // 16) * 3 * 256, 4])
                for z in range(4):     # modes 2 / 3: the `flip` field carries the phase
                    if ci.wp_upf >= 0:
                        pack(ci.w_off, ci.wp_upf + z * ci.wp_ph, ci.cout, ci.cin, 2, (ci.cin * k2, k2, ci.ks, 1), z, mode=2)
                    if ci.wp_s2d >= 0:
                        pack(ci.w_off, ci.wp_s2d + z * ci.wp_ph, ci.cin, ci.cout, 2, (k2, ci.cin * k2, ci.ks, 1), z, mode=3)
        pb.table = np.asarray(recs, dtype=np.int64)
        op = pb.op(_ffi.OP_PACK_BATCH)                 # ONE launch re-packs every conv of the network
        op["i"][0] = len(recs)
        op["i"][1] = (1 if self.bf16 else 0) | (2 if self._x3_packs else 0)      # bit 0: bf16 MFMA fragments (tap pairs) instead of fp32 ones; bit 1: X3 records
        op["l"][0] = max(r[10] for r in recs)
        pb.set_t(op, 0, (S_P, 0))
        pb.set_t(op, 1, (S_WP, 0))
        pb.set_t(op, 2, (S_TAB, 0))
        return pb.finish()

    def ensure_packed(self):
        if not self._packed_ok:
            self._run(self._pack_plan, {S_P: self._flat_data, S_WP: self._wp})
            self._packed_ok = True

    # ---------------------------------------------------------------- plan execution
    def _run(self, plan: Plan, tensors: Dict[int, torch.Tensor], ops: Optional[np.ndarray] = None):
        """ops: run this part of the plan (Plan.main_ops / Plan.tail_ops) instead of the whole op list."""
        if not self._flat_data.is_cuda:
            raise _ffi.CtlError("the HIP engine needs the network on a GPU device; there is no CPU fallback")
        stream = torch.cuda.current_stream()
        if plan.scr_bytes:
            # transient scratch (statistics / reduction partials), reused by consecutive plans: one buffer PER STREAM -- two
            # passes of this network may be in flight on different streams (solver._two_chain_forward)
            if self._scr is None:
                self._scr = {}
            scr = self._scr.get(stream.cuda_stream)
            if scr is None or scr.numel() < plan.scr_bytes:
                scr = self._scr[stream.cuda_stream] = torch.zeros(plan.scr_bytes, dtype=torch.uint8, device=self.device)
            tensors = dict(tensors)
            tensors[S_SCR] = scr
        if plan.table_np is not None:
            if plan.table_dev is None or plan.table_dev.device != self.device:
                plan.table_dev = torch.from_numpy(plan.table_np).to(self.device)
            tensors = dict(tensors)
            tensors[S_TAB] = plan.table_dev
        bases = (ctypes.c_void_p * N_SLOTS)()
        for s, t in tensors.items():
            bases[s] = t.data_ptr()
        run_ops = plan.ops if ops is None else ops
        check(lib.ctl_plan_run(run_ops.ctypes.data, len(run_ops), bases, N_SLOTS, stream.cuda_stream),
              f"{type(self).__name__} plan")

    def set_dropout(self, p: Optional[float]):
        """nn.Dropout2d(p) behind every residual block in training mode (None: off, upstream's default)."""
        if p is not None:
            if not (0.0 <= float(p) < 1.0):
                raise ValueError(f"dropout probability {p!r} must be in [0, 1)")
        self.drop_p = None if p is None else float(p)

    def set_dropout_keep(self, patterns):
        """Test hook: inject the keep patterns of one forward pass ([n, c] tensors of 0/1, one per residual block in forward order;
        None: draw on the device again)."""
        self._drop_keep = None if patterns is None else torch.cat([t.reshape(-1).float() for t in patterns]).to(self.device).contiguous()

    def bn_mode(self) -> str:
        if not self.training:
            return "C"
        return "A" if self._bn_track else "B"

    def _wp_ref(self, off_f):
        return (S_WP, 4 * off_f)

    # ---------------------------------------------------------------- shared block emitters
    def _emit_block_fwd(self, pb: PlanBuilder, prefix: str, pre: str, xin: T, xin_pro, mode: str) -> Tuple[T, dict]:
        """res_convdown (pre='down') / res_up_family (pre='nn' | 'convT'):  out = LReLU(conv1x1(x') + BN(conv(LReLU(BN(conv(x'))))))"""
        C, B = self._convs, self._bns
        train = mode != "C"
        rec = {"prefix": prefix, "pre": pre, "xin": xin, "xin_pro": xin_pro}
        if pre == "down":
            ci = C[prefix + ".down"]
            src, _, _ = pb.conv(xin, self._wp_ref(ci.wp_fwd), ci.cout, 3, stride=2, pro=xin_pro, bias_ref=pb.P(ci.b_off))
            src_mode = _ffi.IN_PLAIN
        elif pre == "convT":
            ci = C[prefix + ".up"]
            src, _, _ = pb.conv(xin, self._wp_ref(ci.wp_fwd), ci.cout, 1, bias_ref=pb.P(ci.b_off), nsub=4)
            src_mode = _ffi.IN_PLAIN
        else:
            src, src_mode = xin, _ffi.IN_UP2
        c0, c3, c1 = C[prefix + ".conv.0"], C[prefix + ".conv.3"], C[prefix + ".conv_input"]
        if pre == "nn" and c0.wp_upf >= 0:
            # conv3x3(nearest_up(x)) as four 2x2 phase convs on x, outputs scattered to (2i+a, 2j+b): 16 taps per 4 outputs, not 36
            u, st, blk = pb.conv(xin, self._wp_ref(c0.wp_upf), c0.cout, 2, nsub=4, pad=2, hout=xin.h, wout=xin.w,
                                 bias_ref=pb.P(c0.b_off), stats=train)
        else:
            u, st, blk = pb.conv(src, self._wp_ref(c0.wp_fwd), c0.cout, 3, in_mode=src_mode, bias_ref=pb.P(c0.b_off), stats=train)
        co1 = pb.bn_forward(B[prefix + ".conv.1"], st, blk, u.n * u.h * u.w, mode)
        v, st, blk = pb.conv(u, self._wp_ref(c3.wp_fwd), c3.cout, 3, pro=(co1["scale"], co1["shift"], SLOPE),
                             bias_ref=pb.P(c3.b_off), stats=train)
        co2 = pb.bn_forward(B[prefix + ".conv.4"], st, blk, v.n * v.h * v.w, mode)
        out, _, _ = pb.conv(src, self._wp_ref(c1.wp_fwd), c1.cout, 1, in_mode=src_mode, bias_ref=pb.P(c1.b_off),
                            res=(v, co2["scale"], co2["shift"]), act=_ffi.ACT_LEAKY, slope=SLOPE)
        rec.update(src=src, src_mode=src_mode, u=u, v=v, out=out, co1=co1, co2=co2)
        if self.drop_p is not None and train:
            # res_x = self.drop(res_x): the block hands on the dropped tensor, keeps `out` (the LeakyReLU derivative needs its sign)
            # and the [n][c] pattern for the backward pass
            keep = pb.act.alloc(4 * out.n * out.c)
            dropped = pb.act.tensor(out.n, out.h, out.w, out.c)
            idx = getattr(pb, "_n_drop", 0)
            pb._n_drop = idx + 1
            keep_in = None
            if self._drop_keep is not None:
                keep_in = (S_KEEP, 4 * getattr(pb, "_keep_off", 0))
                pb._keep_off = getattr(pb, "_keep_off", 0) + out.n * out.c
            pb.dropout(out, dropped, float(self.drop_p), keep_in=keep_in, keep_out=keep, salt=0x5D0 + idx)
            rec["drop"] = (keep, float(self.drop_p))
            return dropped, rec
        return out, rec

    def _tail_of(self, pb: PlanBuilder, rec: Optional[dict]):
        """(T out, T v, slope) of the block whose output gradient the next launch writes, if its tail reduction can ride in that launch."""
        if rec is None or rec.get("drop") is not None or not (FUSE_TAIL16 if pb.b16 else FUSE_TAIL):
            return None
        if pb.b16 and not (rec["out"].b16 and rec["v"].b16):
            return None
        return (rec["out"], rec["v"], SLOPE, rec["pre"] == "nn")

    def _emit_block_bwd(self, pb: PlanBuilder, rec: dict, d_out: T, d_in: Optional[T], need_w: bool, affine: bool, *, pre_tail=None,
                        tail_next=None, act_next=None):
        """Backward of one residual block.  Returns the gradient w.r.t. the block input `xin` (post-activation tensor
        the block consumed; if rec['xin_pro'] is set it is the gradient w.r.t. the *activated* virtual tensor).
        pre_tail = (stats_ref, blocks): `d_out` is already g = dOut * leaky'(out) and the tail's BatchNorm-backward sums are in the
        statistics partials (the launch that wrote it carried CTL_EPI_TAILBWD).  tail_next = `_tail_of` the block that consumes THIS
        block's input gradient: the last launch writing it carries the flag; then returns (d_in, stats_ref, blocks).
        act_next = (T u, scale_ref, shift_ref, slope): the consumer is a conv-BatchNorm-activation pair with BatchNorm input u (`down`
        blocks): the last launch carries CTL_EPI_BNBWD, d_in is g = dAct * act'(BN(u)); same return."""
        assert not (tail_next is not None and act_next is not None)
        ep = dict(tail=tail_next) if act_next is None else dict(bnbwd=act_next, keep_stats=True)
        C, B = self._convs, self._bns
        prefix, pre = rec["prefix"], rec["pre"]
        src, src_mode, u, v, out, xin = rec["src"], rec["src_mode"], rec["u"], rec["v"], rec["out"], rec["xin"]
        c0, c3, c1 = C[prefix + ".conv.0"], C[prefix + ".conv.3"], C[prefix + ".conv_input"]
        A = pb.bscr
        if rec.get("drop") is not None:      # Dropout2d backward: the gradient times the saved pattern (and 1/(1-p))
            d_pre = A.tensor(out.n, out.h, out.w, out.c)
            pb.dropout(d_out, d_pre, rec["drop"][1], keep_in=rec["drop"][0])
            d_out = d_pre
        # bf16: no apply passes at all -- dV and dU stay virtual (g, BatchNorm input, coefficients) and their consumers (the 3x3 weight
        # gradients and the data-gradient convs) evaluate A*g + B*u + C while staging: two launches and three tensor passes less per
        # BatchNorm.  The coefficient tables of a launch sit in LDS: groups * channels <= 256.
        if pb.b16:
            virt = FUSE_BNAPPLY16 and FUSE_BNBWD16 and out.b16 and v.b16 and u.b16 and pb.groups * max(out.c, u.c) <= 256
            # (the block whose input gradient leaves the network as fp32 keeps a stored dU: the staged form writes bf16 only)
            virt_u = virt and not (pre == "nn" and d_in is not None and not d_in.b16)
        else:
            # fp32: the same staging exists (two packed fmas per element next to matrix-bound MFMA loops); dU can only be virtual when its g
            # comes out of the data-gradient epilogue (FUSE_BNBWD)
            virt = FUSE_BNAPPLY and pb.groups * max(out.c, u.c) <= 256 and out.c % 16 == 0 and u.c % 16 == 0
            virt_u = virt and FUSE_BNBWD
        dv2 = du2 = None
        # residual tail: dS (to conv_input) and dV (to conv.3)
        if pre_tail is not None:
            assert rec.get("drop") is None
            ds, dv = d_out, (None if virt else A.tensor(out.n, out.h, out.w, out.c))
            coef2 = pb.bn_backward_from_stats(ds, v, B[prefix + ".conv.4"], rec["co2"], pre_tail[0], pre_tail[1], dx=dv, affine_grad=need_w and affine)
        else:
            ds, dv = A.tensor(out.n, out.h, out.w, out.c), (None if virt else A.tensor(out.n, out.h, out.w, out.c))
            coef2 = pb.bn_backward(0, d_out, out, v, B[prefix + ".conv.4"], rec["co2"], SLOPE, ds=ds, dx=dv, affine_grad=need_w and affine)
        if virt:
            dv, dv2 = ds, (v, coef2)
        pro1 = (rec["co1"]["scale"], rec["co1"]["shift"], SLOPE)
        k9 = 9
        # FUSE_XOUT: the data gradient runs FIRST and leaves the virtual dV behind for the weight gradient
        dv_out = A.tensor(out.n, out.h, out.w, out.c) if (need_w and dv2 is not None and (FUSE_XOUT16 if pb.b16 else FUSE_XOUT)) else None
        def wgrad_c3():
            if need_w:
                if dv_out is not None:
                    pb.wgrad(u, dv_out, 3, pro=pro1, dw_ref=pb.G(c3.w_off), strides=(c3.cin * k9, k9, 3, 1), dbias_ref=pb.G(c3.b_off))
                else:
                    pb.wgrad(u, dv, 3, pro=pro1, dw_ref=pb.G(c3.w_off), strides=(c3.cin * k9, k9, 3, 1), dbias_ref=pb.G(c3.b_off), dy2=dv2)
        if dv_out is None:
            wgrad_c3()
        # dgrad of conv.3; its epilogue already multiplies by leaky'(BN1(u)) and takes the BatchNorm-backward sums (no
        # separate reduction pass); the apply runs in place
        if (FUSE_BNBWD and not pb.b16) or (FUSE_BNBWD16 and pb.b16 and dv.b16 and u.b16):
            g1, st, blk = pb.conv(dv, self._wp_ref(c3.wp_dgrad), c3.cin, 3, arena=A,
                                  bnbwd=(u, rec["co1"]["scale"], rec["co1"]["shift"], SLOPE), x2=dv2, xout=dv_out)
            coef1 = pb.bn_backward_from_stats(g1, u, B[prefix + ".conv.1"], rec["co1"], st, blk, dx=None if virt_u else g1, affine_grad=need_w and affine)
            du = g1
            if virt_u:
                du2 = (u, coef1)
        else:
            da, _, _ = pb.conv(dv, self._wp_ref(c3.wp_dgrad), c3.cin, 3, arena=A, x2=dv2, xout=dv_out)
            # BN1 -> LeakyReLU tail (in place: dU overwrites dA)
            pb.bn_backward(1, da, None, u, B[prefix + ".conv.1"], rec["co1"], SLOPE, ds=None, dx=da, affine_grad=need_w and affine)
            du = da
        if dv_out is not None:
            wgrad_c3()
        ds_low = None
        if pre == "nn":
            # nearest-upsample backward = 2x2 sum-pool; it commutes with pointwise (1x1) convs, so both the 1x1 weight gradient
            # (sum up(x)*dS == sum x*pool(dS)) and the 1x1 data gradient below run on the pooled dS at a quarter of the pixels
            if pre_tail is not None and len(pre_tail) > 2 and pre_tail[2] is not None:
                ds_low = pre_tail[2]            # written by the tail epilogue of the launch that produced dS (FUSE_POOL)
            else:
                ds_low = A.tensor(ds.n, ds.h // 2, ds.w // 2, ds.c)
                pb.sumpool2(ds, ds_low)
        du_out = A.tensor(du.n, du.h, du.w, du.c) if (need_w and du2 is not None and (FUSE_XOUT16 if pb.b16 else FUSE_XOUT)) else None      # (FUSE_XOUT, as for dV above)
        def wgrads_c0_c1():
            if need_w:
                if du_out is not None:
                    pb.wgrad(src, du_out, 3, in_mode=src_mode, dw_ref=pb.G(c0.w_off), strides=(c0.cin * k9, k9, 3, 1), dbias_ref=pb.G(c0.b_off))
                else:
                    pb.wgrad(src, du, 3, in_mode=src_mode, dw_ref=pb.G(c0.w_off), strides=(c0.cin * k9, k9, 3, 1), dbias_ref=pb.G(c0.b_off), dy2=du2)
                if ds_low is not None:
                    pb.wgrad(xin, ds_low, 1, dw_ref=pb.G(c1.w_off), strides=(c1.cin, 1, 1, 1), dbias_ref=pb.G(c1.b_off))
                else:
                    pb.wgrad(src, ds, 1, in_mode=src_mode, dw_ref=pb.G(c1.w_off), strides=(c1.cin, 1, 1, 1), dbias_ref=pb.G(c1.b_off))
        if du_out is None:
            wgrads_c0_c1()
        # gradient w.r.t. x' (full resolution of this block)
        if pre == "down" or pre == "convT":
            dsrc_shape = (src.n, src.h, src.w, src.c)
        else:
            dsrc_shape = (src.n, 2 * src.h, 2 * src.w, src.c)
        if d_in is None:
            d_in = A.tensor(xin.n, xin.h, xin.w, xin.c)
        fin = lambda st, blk: d_in if (tail_next is None and act_next is None) else (d_in, st, blk, pb.last_pool)
        if pre == "nn":
            # sumpool2(conv3x3^T(dU)) as one 4x4 stride-2 conv (no full-resolution gradient tensor at all), then the 1x1 part
            pb.conv(du, self._wp_ref(c0.wp_up), c0.cin, 4, stride=2, out=d_in, x2=du2, xout=du_out)
            if du_out is not None:
                wgrads_c0_c1()
            assert act_next is None
            _, st, blk = pb.conv(ds_low, self._wp_ref(c1.wp_dgrad), c1.cin, 1, out=d_in, accum=True, tail=tail_next)
            return fin(st, blk)
        dsrc = A.tensor(*dsrc_shape)
        pb.conv(du, self._wp_ref(c0.wp_dgrad), c0.cin, 3, out=dsrc, x2=du2, xout=du_out)
        if du_out is not None:
            wgrads_c0_c1()
        pb.conv(ds, self._wp_ref(c1.wp_dgrad), c1.cin, 1, out=dsrc, accum=True)
        if pre == "convT":
            ci = C[prefix + ".up"]
            if need_w:
                pb.chan_sum(dsrc, pb.G(ci.b_off))
                # role swap: "input" = dsrc (full res), "output gradient" = xin  ->  Wt[ci][co][a][b]
                pb.wgrad(dsrc, xin, 2, stride=2, dw_ref=pb.G(ci.w_off), strides=(ci.cout * 4, 4, 2, 1))
            assert act_next is None
            _, st, blk = pb.conv(dsrc, self._wp_ref(ci.wp_dgrad), ci.cin, 2, stride=2, out=d_in, tail=tail_next)
        else:  # down: stride-2 conv
            ci = C[prefix + ".down"]
            if need_w:
                pb.wgrad(xin, dsrc, 3, stride=2, pro=rec["xin_pro"], dw_ref=pb.G(ci.w_off), strides=(ci.cin * k9, k9, 3, 1),
                         dbias_ref=pb.G(ci.b_off))
            if ci.wp_s2d >= 0 and xin.h == 2 * dsrc.h and xin.w == 2 * dsrc.w:
                _, st, blk = pb.conv(dsrc, self._wp_ref(ci.wp_s2d), ci.cin, 2, nsub=4, pad=0, hout=dsrc.h, wout=dsrc.w, out=d_in, **ep)
            else:       # odd sizes: 3x3 conv over the zero-inserted gradient
                _, st, blk = pb.conv(dsrc, self._wp_ref(ci.wp_dgrad), ci.cin, 3, in_mode=_ffi.IN_ZINS2, out=d_in, hout=xin.h, wout=xin.w, **ep)
        return fin(st, blk)

    def _emit_conv_bn_pair_bwd(self, pb, conv_key, bn_key, x: T, x_pro, u: T, co, slope, d_act: T, d_x: Optional[T], need_w, affine,
                               need_dx=True, tail_next=None, pre=None, act_next=None):
        """Backward of  a = act(BN(conv3x3/1x1(x)))  given d_act (gradient w.r.t. a).  Returns gradient w.r.t. x
        (w.r.t. the activated virtual tensor if x_pro is set).
        pre = (stats_ref, blocks): d_act is already g = dAct * act'(BN(u)) and its BatchNorm-backward sums are in the statistics partials
        (the launch that wrote it carried CTL_EPI_BNBWD: `act_next` of its producer); the apply pass then runs inside the consumers' staging
        where they can take it (3x3 weight gradient; 3x3 data gradient into a bf16 tensor).  act_next: see _emit_block_bwd."""
        ci, bn = self._convs[conv_key], self._bns[bn_key]
        A = pb.bscr
        k2 = ci.ks * ci.ks
        dy2 = None
        if pre is not None:
            assert d_act.b16 == u.b16 == pb.b16 and ci.ks == 3 and tail_next is None
            if need_dx and d_x is None:
                d_x = A.tensor(x.n, x.h, x.w, x.c)
            # who cannot stage the apply: in the bf16 family a data gradient into an fp32 or narrow tensor (the staged form writes whole bf16
            # tiles) -- then ONE apply launch writes dU for both consumers (still no reduction pass); the fp32 family's generic epilogue
            # takes any output width (the shape encoder's 4-channel input gradient)
            stored = need_dx and not (d_x.b16 == pb.b16 and (x.c % 16 == 0 or not pb.b16))
            du = A.tensor(u.n, u.h, u.w, u.c) if stored else None
            coef = pb.bn_backward_from_stats(d_act, u, bn, co, pre[0], pre[1], dx=du, affine_grad=need_w and affine)
            if not stored:
                du, dy2 = d_act, (u, coef)
        else:
            du = A.tensor(u.n, u.h, u.w, u.c)
            pb.bn_backward(1, d_act, None, u, bn, co, slope, ds=None, dx=du, affine_grad=need_w and affine)
        du_out = A.tensor(u.n, u.h, u.w, u.c) if (need_w and need_dx and dy2 is not None and (FUSE_XOUT16 if pb.b16 else FUSE_XOUT)) else None      # (FUSE_XOUT, see _emit_block_bwd)
        def wgrad():
            if need_w:
                pb.wgrad(x, du if du_out is None else du_out, ci.ks, pro=x_pro, in_mode=_ffi.IN_C4 if ci.wp_c4 >= 0 else 0, dw_ref=pb.G(ci.w_off),
                         strides=(ci.cin * k2, k2, ci.ks, 1), dbias_ref=pb.G(ci.b_off), dy2=dy2 if du_out is None else None)
        if du_out is None:
            wgrad()
        if not need_dx:
            return None
        if d_x is None:
            d_x = A.tensor(x.n, x.h, x.w, x.c)
        ep = dict(tail=tail_next) if act_next is None else dict(bnbwd=act_next, keep_stats=True)
        _, st, blk = pb.conv(du, self._wp_ref(ci.wp_dgrad), ci.cin, ci.ks, out=d_x, x2=dy2, xout=du_out, **ep)
        if du_out is not None:
            wgrad()
        return d_x if (tail_next is None and act_next is None) else (d_x, st, blk, pb.last_pool)

    # ---------------------------------------------------------------- public compute entry points
    def _alloc_out(self, shape):
        n, h, w, c = shape
        return torch.empty((n, c, h, w), dtype=torch.float32, device=self.device, memory_format=torch.channels_last)

    def run_forward(self, x: torch.Tensor, mode: str, groups: int = 1, stack: Optional[PassStack] = None):
        """Returns (outputs tuple, act workspace tensor, plan).  x: logical NCHW, NHWC memory.  groups > 1: x stacks that many
        independent batches along n; BatchNorm treats each on its own (statistics, running-stat updates in order).
        stack: the pass becomes the next slot of that PassStack (shared N-stacked arena and outputs, see PassStack)."""
        n, c, h, w = x.shape
        if c != self.cin:
            raise ValueError(f"{type(self).__name__}: expected {self.cin} input channels, got {c}")
        if groups < 1 or n % groups:
            raise ValueError(f"{type(self).__name__}: batch {n} cannot be split into {groups} groups")
        pp = (0, 1) if stack is None else (stack.filled, stack.P)
        key = ("f", n, h, w, mode, groups, self.drop_p, self._drop_keep is not None) + (() if stack is None else (pp,))
        plan = self._plans.get(key)
        if plan is None:
            self._cur_groups, self._cur_pp = groups, pp
            try:
                plan = self._plans[key] = self._compile_forward(n, h, w, mode)
            finally:
                self._cur_groups, self._cur_pp = 1, (0, 1)
        self.ensure_packed()
        if stack is None:
            act = self._arenas.acquire(plan.act_bytes, self.device)        # lease: lives as long as the autograd context of this pass
            outs = [self._alloc_out(s) for s in plan.out_shapes]
        else:
            if stack.lease is None:
                if mode == "C" or self.drop_p is not None:
                    raise _ffi.CtlError("PassStack: training-mode passes of a network without Dropout2d only")
                stack.lease = self._arenas.acquire(plan.act_bytes, self.device)
                stack.outs = [self._alloc_out((s[0] * stack.P,) + tuple(s[1:])) for s in plan.out_shapes]
                stack.shape = (n, h, w, groups)
            act = stack.lease
            outs = [o[pp[0] * n:(pp[0] + 1) * n] for o in stack.outs]
        tensors = {S_X: x, S_P: self._flat_data, S_B: self._bflat, S_NBT: self._nbt, S_WP: self._wp, S_ACT: act.t, S_OUT0: outs[0]}
        if len(outs) > 1:
            tensors[S_OUT1] = outs[1]
        if self.drop_p is not None and mode != "C":
            if self._drop_keep is not None:
                tensors[S_KEEP] = self._drop_keep
            else:
                if self._drop_state is None:      # (drawn at first use: the construction-time RNG order of the weight init stays upstream's)
                    from .model_util import _draw_seed
                    self._drop_state = torch.tensor([_draw_seed() & (2 ** 62 - 1), 0, 0], dtype=torch.int64, device=self.device)
                check(lib.ctl_step_tick(self._drop_state.data_ptr(), torch.cuda.current_stream().cuda_stream), "ctl_step_tick")
                tensors[S_STATE] = self._drop_state
        self._run(plan, tensors)
        if stack is not None:
            stack.plans.append(plan)
            stack.modes.append(mode)
            stack.xs.append(x)
            stack.douts.append([None] * len(outs))
            stack.events.append([])
        return tuple(outs), act, plan

    # ---------------------------------------------------------------- stacked backward (PassStack)
    @staticmethod
    def _gather_stacked(parts, alloc):
        """[P * n, ...] tensor whose p-th block is parts[p] (None = zeros): the parts themselves when they already are consecutive
        blocks of one allocation (views of a stacked output / gradient), else one copy each into a fresh tensor."""
        ref = next(t for t in parts if t is not None)
        n = ref.shape[0]
        nbytes = ref.numel() * ref.element_size()
        if all(t is not None and t.shape == ref.shape and t.dtype == ref.dtype and t.is_contiguous(memory_format=torch.channels_last)
               and t.data_ptr() == parts[0].data_ptr() + k * nbytes for k, t in enumerate(parts)):
            base = parts[0]._base if parts[0]._base is not None else None
            if base is not None and base.dim() == 4 and base.shape[1:] == ref.shape[1:] and base.is_contiguous(memory_format=torch.channels_last):
                k0 = (parts[0].data_ptr() - base.data_ptr()) // nbytes
                if base.data_ptr() + k0 * nbytes == parts[0].data_ptr() and (k0 + len(parts)) * n <= base.shape[0]:
                    return base[k0 * n:(k0 + len(parts)) * n]
        out = alloc((n * len(parts), ref.shape[2], ref.shape[3], ref.shape[1]))
        for k, t in enumerate(parts):
            if t is None:
                out[k * n:(k + 1) * n].zero_()
            else:
                out[k * n:(k + 1) * n].copy_(t)
        return out

    def backward_stack(self, stack: PassStack, tail_stream: Optional["torch.cuda.Stream"] = None):
        """Issue the backward of the passes of `stack` on the current stream (which first waits for the gradients the passes received
        on other streams): ONE launch chain over the stacked batch when every slot is filled, else one per filled slot.  The flat
        parameter gradient is parked like a per-pass gradient (collect_deferred_grads).  Returns [(root input tensor, its gradient)]
        for the passes whose input wants a gradient: the caller hands them back to autograd.
        tail_stream: the weight-gradient family of the stacked plan runs there, behind the data-gradient chain on the current stream
        (the chain the NEXT network's backward depends on): the fat weight-gradient launches fill the second chain."""
        assert not stack.done
        stack.done = True
        cur = torch.cuda.current_stream()
        for evs in stack.events:
            for ev in evs:
                cur.wait_event(ev)
        P, n = stack.P, (stack.shape[0] if stack.shape else 0)
        cont = []
        live = [p for p in range(stack.filled) if any(d is not None for d in stack.douts[p])]
        if not live:
            return cont
        need_w = True
        if stack.filled == P and len(live) == P:
            n_out = len(stack.outs)
            douts = []
            for k in range(n_out):
                parts = [stack.douts[p][k] for p in range(P)]
                douts.append(None if all(t is None for t in parts) else self._gather_stacked(parts, self._alloc_out))
            x = self._gather_stacked(list(stack.xs), self._alloc_out)
            need_dx = any(stack.need_dx)
            dx, gflat = self._run_backward_stacked(stack, x, tuple(douts), need_dx, need_w, tail_stream)
            if need_dx:
                cont = [(stack.roots[p], dx[p * n:(p + 1) * n]) for p in range(P) if stack.need_dx[p]]
            with torch.cuda.stream(tail_stream if tail_stream is not None else cur):
                self._park_grad(stack.seqs[0], gflat)
        else:
            for p in live:
                outs = tuple(o[p * n:(p + 1) * n] for o in stack.outs)
                dx, gflat = self.run_backward(stack.xs[p], stack.lease, outs, stack.plans[p], stack.modes[p], tuple(stack.douts[p]),
                                              stack.need_dx[p], need_w, stack.modes[p] == "A")
                if stack.need_dx[p]:
                    cont.append((stack.roots[p], dx))
                self._park_grad(stack.seqs[p], gflat)
        if self._pending_bwd == 0 and self._on_grads_complete is not None and self._defer_grads:
            with torch.cuda.stream(tail_stream if (tail_stream is not None and stack.filled == P and len(live) == P) else cur):
                self._on_grads_complete()
        return cont

    def _park_grad(self, seq, gflat):
        if gflat is None:
            return
        if self._defer_grads:
            ev = torch.cuda.Event()
            ev.record()
            self._deferred.append((seq, gflat, ev))
        else:
            self._flat.grad.add_(gflat)
            self._grad_written, self._grad_is_zero = True, False

    def _run_backward_stacked(self, stack: PassStack, x, douts, need_dx: bool, need_w: bool, tail_stream=None):
        P = stack.P
        n, h, w, G = stack.shape
        plan0 = stack.plans[0]
        mask = tuple(d is not None for d in douts)
        full = (1 << (P * G)) - 1
        amask = sum((((1 << G) - 1) << (p * G)) for p in range(P) if stack.modes[p] == "A")
        key = ("bs", n, h, w, tuple(stack.modes), mask, need_dx, need_w, P, G, id(plan0))
        plan = self._plans.get(key)
        if plan is None:
            view = _StackedForward(_stack_rec(plan0.rec, P), P * G)
            self._cur_groups, self._cur_affine_mask, self._cur_split = P * G, (0 if amask == full else amask), True
            try:
                plan = self._plans[key] = self._compile_backward(view, "A", mask, need_dx, need_w, amask != 0)
            finally:
                self._cur_groups, self._cur_affine_mask, self._cur_split = 1, 0, False
        lease = self._arenas.acquire(plan.bscr_bytes, self.device)
        stack.keep.append(lease)          # (the tail may still read the arena on another stream when this call returns: held until the step ends)
        self._dbg_last = (plan, lease.t)
        tensors = {S_X: x, S_P: self._flat_data, S_WP: self._wp, S_ACT: stack.lease.t, S_BSCR: lease.t, S_OUT0: stack.outs[0]}
        if len(stack.outs) > 1:
            tensors[S_OUT1] = stack.outs[1]
        for slot, d in zip((S_DOUT0, S_DOUT1), douts):
            if d is not None:
                tensors[slot] = d
        dx = gflat = None
        if need_dx:
            dx = self._alloc_out((P * n, h, w, self.cin))
            tensors[S_DX] = dx
        if need_w:
            gflat = torch.empty(self._pcount, dtype=torch.float32, device=self.device)
            tensors[S_GRAD] = gflat
            self._grad_written, self._grad_is_zero = True, False
        if tail_stream is None or plan.tail_ops is None or not need_w:
            self._run(plan, tensors)
            return dx, gflat
        self._run(plan, tensors, ops=plan.main_ops)
        done = torch.cuda.Event()
        done.record()
        with torch.cuda.stream(tail_stream):
            tail_stream.wait_event(done)
            for t in tensors.values():
                t.record_stream(tail_stream)
            self._run(plan, tensors, ops=plan.tail_ops)
        return dx, gflat

    def run_backward(self, x, act, outs, fwd_plan: Plan, mode: str, douts, need_dx: bool, need_w: bool, affine: bool):
        """Returns (dx or None, flat parameter gradient or None)."""
        n, c, h, w = x.shape
        mask = tuple(d is not None for d in douts)
        key = ("b", n, h, w, mode, mask, need_dx, need_w, affine, fwd_plan.groups, id(fwd_plan))
        plan = self._plans.get(key)
        if plan is None:
            self._cur_groups = fwd_plan.groups
            try:
                plan = self._plans[key] = self._compile_backward(fwd_plan, mode, mask, need_dx, need_w, affine)
            finally:
                self._cur_groups = 1
        lease = self._arenas.acquire(plan.bscr_bytes, self.device)
        bscr = lease.t
        self._dbg_last = (plan, bscr)          # lets tests inspect intermediate gradients (valid until the next pass on this stream:
                                               # the lease ends with this call and the arena is then reused in launch order)
        tensors = {S_X: x, S_P: self._flat_data, S_WP: self._wp, S_ACT: act.t, S_BSCR: bscr, S_OUT0: outs[0]}
        if len(outs) > 1:
            tensors[S_OUT1] = outs[1]
        for slot, d in zip((S_DOUT0, S_DOUT1), douts):
            if d is not None:
                tensors[slot] = d
        dx = gflat = None
        if need_dx:
            dx = self._alloc_out((n, h, w, c))
            tensors[S_DX] = dx
        if need_w:
            gflat = torch.empty(self._pcount, dtype=torch.float32, device=self.device)
            tensors[S_GRAD] = gflat
            self._grad_written = True
            self._grad_is_zero = False     # (the pass's gradient reaches the buffer through autograd or collect_deferred_grads)
        self._run(plan, tensors)
        return dx, gflat


# ================================================================================================ encoders
class MyEncoder(CtlNet):
    """MyEncoder(feature_reduce=4, norm=BatchNorm2d, act=ReLU): ladder 16-32-64-128-128, spatial /16."""

    def __init__(self, input_channel: int, feature_reduce: int = 4, device="cuda", _spec=None, _prefix="", bf16: bool = False):
        self._px = _prefix
        self.chan = [64 // feature_reduce, 128 // feature_reduce, 256 // feature_reduce, 512 // feature_reduce,
                     512 // feature_reduce]
        super().__init__(_spec if _spec is not None else _init.encoder_spec("", input_channel, feature_reduce),
                         input_channel, device, bf16)

    # -- plan pieces shared with the dual encoder
    def _emit_encoder_fwd(self, pb: PlanBuilder, x: T, mode: str, z_out: T):
        C, B, px = self._convs, self._bns, self._px
        train = mode != "C"
        c0, c3 = C[px + "inc.0"], C[px + "inc.3"]
        if c0.wp_c4 >= 0:
            u0, st, blk = pb.conv(x, self._wp_ref(c0.wp_c4), c0.cout, 3, in_mode=_ffi.IN_C4, bias_ref=pb.P(c0.b_off), stats=train)
        else:
            u0, st, blk = pb.conv(x, self._wp_ref(c0.wp_fwd), c0.cout, 3, bias_ref=pb.P(c0.b_off), stats=train)
        co0 = pb.bn_forward(B[px + "inc.1"], st, blk, u0.n * u0.h * u0.w, mode)
        v0, st, blk = pb.conv(u0, self._wp_ref(c3.wp_fwd), c3.cout, 3, pro=(co0["scale"], co0["shift"], SLOPE),
                              bias_ref=pb.P(c3.b_off), stats=train)
        co1 = pb.bn_forward(B[px + "inc.4"], st, blk, v0.n * v0.h * v0.w, mode)
        rec = {"x": x, "u0": u0, "v0": v0, "co0": co0, "co1": co1, "blocks": []}
        cur, cur_pro = v0, (co1["scale"], co1["shift"], SLOPE)      # x1 = LReLU(BN(v0)) stays virtual
        for i in range(1, 5):
            cur, brec = self._emit_block_fwd(pb, f"{px}down{i}", "down", cur, cur_pro, mode)
            cur_pro = None
            rec["blocks"].append(brec)
        cf = C[px + "final_conv.0"]
        uf, st, blk = pb.conv(cur, self._wp_ref(cf.wp_fwd), cf.cout, 1, bias_ref=pb.P(cf.b_off), stats=train)
        cof = pb.bn_forward(B[px + "final_conv.1"], st, blk, uf.n * uf.h * uf.w, mode)
        pb.bn_act(uf, cof, 0.0, z_out)                               # act = ReLU
        rec.update(uf=uf, cof=cof, z=z_out, x4=cur)
        return rec

    def _emit_encoder_bwd(self, pb: PlanBuilder, rec, dz: T, need_dx: bool, need_w: bool, affine: bool):
        px = self._px
        dbg = {"dz": dz}                 # where the intermediate gradients live (tests read them through CtlNet._dbg_last)
        blocks = rec["blocks"]
        tail = self._tail_of(pb, blocks[-1])
        d = self._emit_conv_bn_pair_bwd(pb, px + "final_conv.0", px + "final_conv.1", rec["x4"], None, rec["uf"], rec["cof"], 0.0,
                                        dz, None, need_w, affine, tail_next=tail)
        pre = None
        if tail is not None:
            d, pre = d[0], d[1:]
        dbg["d_down5"] = d               # gradient w.r.t. the output of down4 (with FUSE_TAIL: already times leaky'(out), i.e. dS of down4)
        dbg["tail_down5"] = pre          # ... and where the tail's BatchNorm-backward sums of down4 are (statistics partials ref, rows)
        u0, v0 = rec["u0"], rec["v0"]
        if pb.b16:
            pair16 = FUSE_PAIR16 and FUSE_BNAPPLY16 and u0.b16 and v0.b16
        else:
            pair16 = FUSE_PAIR and FUSE_BNAPPLY and FUSE_BNBWD and u0.c % 16 == 0
        pair16 = pair16 and pb.groups * u0.c <= 256 and blocks[0].get("drop") is None
        act1 = (v0, rec["co1"]["scale"], rec["co1"]["shift"], SLOPE) if pair16 else None
        act0 = (u0, rec["co0"]["scale"], rec["co0"]["shift"], SLOPE) if pair16 else None
        for i, brec in reversed(list(enumerate(blocks))):
            tail = self._tail_of(pb, blocks[i - 1]) if i >= 1 else None
            d = self._emit_block_bwd(pb, brec, d, None, need_w, affine, pre_tail=pre, tail_next=tail, act_next=act1 if i == 0 else None)
            pre = None
            if tail is not None or (i == 0 and act1 is not None):
                d, pre = d[0], d[1:]
            dbg[f"d_down{i + 1}"] = d    # gradient w.r.t. the input of block down{i+1} (down1 with FUSE_PAIR16: g of the inc.3 pair)
            dbg[f"tail_down{i + 1}"] = pre
        # d = gradient w.r.t. x1 = LReLU(BN(v0))
        pro0 = (rec["co0"]["scale"], rec["co0"]["shift"], SLOPE)
        d = self._emit_conv_bn_pair_bwd(pb, px + "inc.3", px + "inc.4", u0, pro0, v0, rec["co1"], SLOPE, d, None,
                                        need_w, affine, pre=pre, act_next=act0)
        pre = None
        if act0 is not None:
            d, pre = d[0], d[1:]
        dbg["d_inc3"], dbg["pre_inc3"] = d, pre
        dx = T((S_DX, 0), *rec["x"][1:]) if need_dx else None
        self._emit_conv_bn_pair_bwd(pb, px + "inc.0", px + "inc.1", rec["x"], None, u0, rec["co0"], SLOPE, d, dx,
                                    need_w, affine, need_dx=need_dx, pre=pre)
        return dbg

    def _zdims(self, n, h, w):
        for _ in range(4):
            h, w = (h + 1) // 2, (w + 1) // 2
        return (n, h, w, self.chan[4])

    def _compile_forward(self, n, h, w, mode) -> Plan:
        pb = PlanBuilder(self)
        zs = self._zdims(n, h, w)
        rec = self._emit_encoder_fwd(pb, T((S_X, 0), n, h, w, self.cin), mode, T((S_OUT0, 0), *zs))
        return pb.finish(rec, [zs])

    def _compile_backward(self, fwd: Plan, mode, mask, need_dx, need_w, affine) -> Plan:
        pb = PlanBuilder(self)
        if need_w:
            pb.zero((S_GRAD, 0), 4 * self._pcount)
        z = fwd.rec["z"]
        dbg = self._emit_encoder_bwd(pb, fwd.rec, T((S_DOUT0, 0), *z[1:]), need_dx, need_w, affine)
        return pb.finish(dbg)

    def forward(self, x):
        from .autograd import net_apply
        return net_apply(self, x)[0]


class Dual_Branch_Encoder(MyEncoder):
    """FTN encoder: z_i = general_encoder(x), z_s = code_decoupler(z_i)."""

    def __init__(self, input_channel: int, z_level_1_channel: int = 128, z_level_2_channel: int = 128, feature_reduce: int = 4,
                 device="cuda", bf16: bool = False):
        assert z_level_1_channel == z_level_2_channel == 512 // feature_reduce
        super().__init__(input_channel, feature_reduce, device,
                         _spec=_init.dual_encoder_spec(input_channel, z_level_1_channel, feature_reduce),
                         _prefix="general_encoder.", bf16=bf16)

    def _compile_forward(self, n, h, w, mode) -> Plan:
        pb = PlanBuilder(self)
        C, B = self._convs, self._bns
        train = mode != "C"
        zs = self._zdims(n, h, w)
        z_i, z_s = T((S_OUT0, 0), *zs), T((S_OUT1, 0), *zs)
        rec = self._emit_encoder_fwd(pb, T((S_X, 0), n, h, w, self.cin), mode, z_i)
        d0, d3 = C["code_decoupler.0"], C["code_decoupler.3"]
        ud, st, blk = pb.conv(z_i, self._wp_ref(d0.wp_fwd), d0.cout, 3, bias_ref=pb.P(d0.b_off), stats=train)
        cod0 = pb.bn_forward(B["code_decoupler.1"], st, blk, ud.n * ud.h * ud.w, mode)
        vd, st, blk = pb.conv(ud, self._wp_ref(d3.wp_fwd), d3.cout, 3, pro=(cod0["scale"], cod0["shift"], SLOPE),
                              bias_ref=pb.P(d3.b_off), stats=train)
        cod1 = pb.bn_forward(B["code_decoupler.4"], st, blk, vd.n * vd.h * vd.w, mode)
        pb.bn_act(vd, cod1, 0.0, z_s)                                # nn.ReLU at the end of code_decoupler
        rec.update(ud=ud, vd=vd, cod0=cod0, cod1=cod1, z_s=z_s)
        return pb.finish(rec, [zs, zs])

    def _compile_backward(self, fwd: Plan, mode, mask, need_dx, need_w, affine) -> Plan:
        pb = PlanBuilder(self)
        rec = fwd.rec
        if need_w:
            pb.zero((S_GRAD, 0), 4 * self._pcount)
        z_i = rec["z"]
        dzi_in = T((S_DOUT0, 0), *z_i[1:]) if mask[0] else None
        extra = {}
        if mask[1]:
            dzs = T((S_DOUT1, 0), *z_i[1:])
            pro = (rec["cod0"]["scale"], rec["cod0"]["shift"], SLOPE)
            d = self._emit_conv_bn_pair_bwd(pb, "code_decoupler.3", "code_decoupler.4", rec["ud"], pro, rec["vd"], rec["cod1"], 0.0,
                                            dzs, None, need_w, affine)
            # conv d.0 consumes z_i: its dgrad is added to the gradient arriving at z_i directly
            ci, bn = self._convs["code_decoupler.0"], self._bns["code_decoupler.1"]
            du = pb.bscr.tensor(*rec["ud"][1:])
            pb.bn_backward(1, d, None, rec["ud"], bn, rec["cod0"], SLOPE, ds=None, dx=du, affine_grad=need_w and affine)
            if need_w:
                pb.wgrad(z_i, du, 3, dw_ref=pb.G(ci.w_off), strides=(ci.cin * 9, 9, 3, 1), dbias_ref=pb.G(ci.b_off))
            dzi = pb.bscr.tensor(*z_i[1:], b16=False)       # (sums the fp32 gradient arriving at z_i directly: fp32, it is 2 MiB)
            if dzi_in is not None:      # dz_i = (gradient arriving at z_i directly) + dgrad of code_decoupler.0
                pb.copy(dzi_in.ref, dzi.ref, 4 * z_i.n * z_i.h * z_i.w * z_i.c)
            pb.conv(du, self._wp_ref(ci.wp_dgrad), ci.cin, 3, out=dzi, accum=dzi_in is not None)
            extra = {"d_cd3": d, "du_cd0": du}
        else:
            dzi = dzi_in
        if dzi is None:
            raise _ffi.CtlError("Dual_Branch_Encoder backward called without any output gradient")
        dbg = self._emit_encoder_bwd(pb, rec, dzi, need_dx, need_w, affine)
        dbg.update(extra)
        return pb.finish(dbg)

    def forward(self, x):
        from .autograd import net_apply
        return net_apply(self, x)

    def _compile_filter(self, n, h, w, mode) -> Plan:
        """code_decoupler alone (encoder_decoder.py:496-498): z_i [n, h, w, C] in the input slot -> z_s."""
        pb = PlanBuilder(self)
        C, B = self._convs, self._bns
        train = mode != "C"
        d0, d3 = C["code_decoupler.0"], C["code_decoupler.3"]
        z_i, z_s = T((S_X, 0), n, h, w, d0.cin), T((S_OUT0, 0), n, h, w, d3.cout)
        ud, st, blk = pb.conv(z_i, self._wp_ref(d0.wp_fwd), d0.cout, 3, bias_ref=pb.P(d0.b_off), stats=train)
        cod0 = pb.bn_forward(B["code_decoupler.1"], st, blk, ud.n * ud.h * ud.w, mode)
        vd, st, blk = pb.conv(ud, self._wp_ref(d3.wp_fwd), d3.cout, 3, pro=(cod0["scale"], cod0["shift"], SLOPE),
                              bias_ref=pb.P(d3.b_off), stats=train)
        cod1 = pb.bn_forward(B["code_decoupler.4"], st, blk, vd.n * vd.h * vd.w, mode)
        pb.bn_act(vd, cod1, 0.0, z_s)
        return pb.finish({}, [(n, h, w, d3.cout)])

    def filter_code(self, z):
        """z_s = code_decoupler(z_i) on its own, in the network's current BatchNorm mode.  Forward only: upstream never calls it
        outside `forward` (advanced_triplet_recon_segmentation_model.py:208-221 is its one, unused, caller), so no backward plan
        exists for it; a `z` that requires grad is refused instead of silently cutting the graph."""
        from . import ops
        ops.require_gpu(z)
        if torch.is_grad_enabled() and z.requires_grad:
            raise _ffi.CtlError("filter_code: forward-only entry (call the encoder's forward for a differentiable z_s)")
        z = ops.as_nhwc(z.detach().float())
        n, c, h, w = z.shape
        d0 = self._convs["code_decoupler.0"]
        if c != d0.cin:
            raise ValueError(f"filter_code: expected {d0.cin} channels, got {c}")
        mode = self.bn_mode()
        key = ("filter", n, h, w, mode)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._compile_filter(n, h, w, mode)
        self.ensure_packed()
        act = self._arenas.acquire(plan.act_bytes, self.device)
        out = self._alloc_out(plan.out_shapes[0])
        self._run(plan, {S_X: z, S_P: self._flat_data, S_B: self._bflat, S_NBT: self._nbt, S_WP: self._wp, S_ACT: act.t, S_OUT0: out})
        return out


# ================================================================================================ decoder
class MyDecoder(CtlNet):
    """MyDecoder(up_type in {'NN','Conv2'}, norm=BatchNorm2d, last_act in {None, Sigmoid})."""

    def __init__(self, input_channel: int, output_channel: int, feature_reduce: int = 4, up_type: str = "NN",
                 last_act: Optional[str] = None, device="cuda", bf16: bool = False):
        if up_type not in ("NN", "Conv2"):
            raise NotImplementedError(f"up_type {up_type!r} (the reference's FCN_16_standard uses 'NN' and 'Conv2')")
        self.up_type, self.out_ch = up_type, output_channel
        self.sigmoid = last_act in ("sigmoid", "Sigmoid") or isinstance(last_act, nn.Sigmoid)
        super().__init__(_init.decoder_spec(input_channel, output_channel, up_type, feature_reduce), input_channel, device, bf16)

    def _compile_forward(self, n, h, w, mode) -> Plan:
        pb = PlanBuilder(self)
        cur = T((S_X, 0), n, h, w, self.cin)
        rec = {"x": cur, "blocks": []}
        pre = "nn" if self.up_type == "NN" else "convT"
        for i in range(1, 5):
            cur, brec = self._emit_block_fwd(pb, f"up{i}", pre, cur, None, mode)
            rec["blocks"].append(brec)
        cf = self._convs["final_conv"]
        out = T((S_OUT0, 0), cur.n, cur.h, cur.w, cf.cout)
        pb.conv(cur, self._wp_ref(cf.wp_fwd), cf.cout, 1, bias_ref=pb.P(cf.b_off), act=_ffi.ACT_SIGMOID if self.sigmoid else 0,
                out=out)
        rec.update(x4=cur, out=out)
        return pb.finish(rec, [(out.n, out.h, out.w, out.c)])

    def _compile_backward(self, fwd: Plan, mode, mask, need_dx, need_w, affine) -> Plan:
        pb = PlanBuilder(self)
        rec = fwd.rec
        if need_w:
            pb.zero((S_GRAD, 0), 4 * self._pcount)
        out, x4, cf = rec["out"], rec["x4"], self._convs["final_conv"]
        dout = T((S_DOUT0, 0), *out[1:])
        if self.sigmoid:
            dl = pb.bscr.tensor(*out[1:], b16=False)
            pb.sigmoid_bwd(dout, out, dl)
            dout = dl
        if need_w:
            pb.wgrad(x4, dout, 1, dw_ref=pb.G(cf.w_off), strides=(cf.cin, 1, 1, 1), dbias_ref=pb.G(cf.b_off))
        blocks = rec["blocks"]
        tail = self._tail_of(pb, blocks[3])
        d, st, blk = pb.conv(dout, self._wp_ref(cf.wp_dgrad), cf.cin, 1, arena=pb.bscr, tail=tail)
        pre = (st, blk, pb.last_pool) if tail is not None else None
        dbg = {"d_out4": d, "tail_out4": pre}
        for i in range(3, -1, -1):
            d_in = T((S_DX, 0), *rec["x"][1:]) if (i == 0 and need_dx) else None
            tail = self._tail_of(pb, blocks[i - 1]) if i >= 1 else None
            d = self._emit_block_bwd(pb, blocks[i], d, d_in, need_w, affine, pre_tail=pre, tail_next=tail)
            pre = None
            if tail is not None:
                d, pre = d[0], d[1:]
            dbg[f"d_out{i}"] = d
            dbg[f"tail_out{i}"] = pre
        return pb.finish(dbg)

    def forward(self, x):
        from .autograd import net_apply
        return net_apply(self, x)[0]


def build_networks(image_ch: int = 1, num_classes: int = 4, reduce_factor: int = 4, device="cuda",
                   state_dicts: Optional[dict] = None, dtype: str = "fp32", encoder_dropout: Optional[float] = None,
                   decoder_dropout: Optional[float] = None) -> Dict[str, CtlNet]:
    """`get_network('FCN_16_standard')` (model.py:76-149).  Without `state_dicts` the weights are drawn exactly like the
    reference does for the current torch seed (see init.py)."""
    z = 512 // reduce_factor
    sds = state_dicts if state_dicts is not None else _init.reference_init_state_dicts(image_ch, num_classes, reduce_factor)
    if dtype not in ("fp32", "bf16"):
        raise ValueError(f"dtype {dtype!r}: 'fp32' (the reference's arithmetic) or 'bf16' (bf16 storage + MFMA, fp32 accumulate)")
    b = dtype == "bf16"
    nets = {
        "image_encoder": Dual_Branch_Encoder(image_ch, z, z, reduce_factor, device=device, bf16=b),
        "segmentation_decoder": MyDecoder(z, num_classes, reduce_factor, "NN", None, device=device, bf16=b),
        "shape_encoder": MyEncoder(num_classes, reduce_factor, device=device, bf16=b),
        "shape_decoder": MyDecoder(z, num_classes, reduce_factor, "NN", None, device=device, bf16=b),
        "image_decoder": MyDecoder(z, image_ch, reduce_factor, "Conv2", "sigmoid", device=device, bf16=b),
    }
    for k, net in nets.items():
        if k in sds and sds[k] is not None:
            net.load_state_dict(sds[k])
        # model.py:92-106: encoder_dropout -> image_encoder, shape_encoder; decoder_dropout -> the three decoders
        net.set_dropout(encoder_dropout if k.endswith("encoder") else decoder_dropout)
    return nets
