"""A captured two-chain hipGraph replayed as LINEAR segment graphs on two streams.

The whole-step graph (graph.py) has two launch chains that fork and join a dozen times (bf16 step: 932 nodes, 13 forks, 13 joins, never
wider than two).  The HIP runtime replays a linear graph on a fast path (0.3-0.6 ms of host time for ~900 nodes) but a graph with
branches node by node through its general path: ~1 us more per node, which the short kernels of the bf16 step do not hide (replay 10.35 ms
against 9.4-9.5 ms for the same launches issued live on two streams).  `SegmentReplay` takes the captured
`hipGraph_t`, covers it with chains (a node continues the chain of a predecessor whose last node it follows), cuts every chain where an edge
crosses to or from another chain, builds one linear graph per segment (a clone of the captured graph with every other node removed:
no kernel parameters are touched) and replays them in topological order: segment launches on the chain's stream, one event per segment that
another chain waits for.  Same nodes, same edges (a cross-chain edge becomes record -> wait), so the same results bit for bit
(tests/test_graph_gpu.py); 22 launches + 11 events cost the host 0.8 ms per step.

This is stream / graph plumbing over the HIP runtime API (ctypes on the libamdhip64.so PyTorch has already loaded); no kernels here."""
from __future__ import annotations

import ctypes as C
import heapq
import sys
from typing import List

import torch

from ._ffi import CtlError

_VP = C.c_void_p
_hip = None
_SIGS = {
    "hipGraphGetNodes": [_VP, _VP, _VP], "hipGraphGetEdges": [_VP, _VP, _VP, _VP], "hipGraphClone": [_VP, _VP],
    "hipGraphNodeFindInClone": [_VP, _VP, _VP], "hipGraphDestroyNode": [_VP], "hipGraphInstantiate": [_VP, _VP, _VP, _VP, C.c_size_t],
    "hipGraphLaunch": [_VP, _VP], "hipGraphExecDestroy": [_VP], "hipGraphDestroy": [_VP], "hipEventCreateWithFlags": [_VP, C.c_uint],
    "hipEventDestroy": [_VP], "hipEventRecord": [_VP, _VP], "hipStreamWaitEvent": [_VP, _VP, C.c_uint],
}
_EVENT_DISABLE_TIMING = 0x2
MAX_CHAINS = 4


def _lib():
    global _hip
    if _hip is None:
        try:
            lib = C.CDLL("libamdhip64.so")
        except OSError as exc:
            raise CtlError(f"SegmentReplay needs the HIP runtime library: {exc}")
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, C.c_int
        _hip = lib
    return _hip


def _ck(rc: int, what: str):
    if rc != 0:
        raise CtlError(f"SegmentReplay: {what} failed with HIP error {rc}")


def streams_overlap_ratio(a: "torch.cuda.Stream", b: "torch.cuda.Stream", us: int = 300) -> float:
    """Two idle `us`-microsecond kernels (ctl_spin), one per stream, between two events: ~1.0 = they ran side by side, ~2.0 = one behind
    the other.  HIP streams share a few hardware queues (4 by default) in creation order, and two streams on one queue run IN ORDER."""
    from ._ffi import lib, check
    for st in (a, b):
        check(lib.ctl_spin(1, st.cuda_stream), "ctl_spin")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.wait_stream(a)
    e0.record(a)
    for st in (a, b):
        check(lib.ctl_spin(us, st.cuda_stream), "ctl_spin")
    a.wait_stream(b)
    e1.record(a)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / us


class _Segment:
    __slots__ = ("chain", "nodes", "waits", "event", "exe", "graph")

    def __init__(self, chain):
        self.chain, self.nodes, self.waits, self.event, self.exe, self.graph = chain, [], [], None, None, None


def plan_segments(n_nodes: int, edges, emit: str = "close") -> "tuple[List[int], List[_Segment]]":
    """Pure host logic (no HIP): nodes 0..n-1 in creation order, edges (a, b) = b depends on a.  Returns (chain of every node, the segments
    in launch order).  Every node is in exactly one segment; inside a segment the nodes are consecutive nodes of one chain; an edge
    between two chains is covered by `waits` (the consumer's segment waits for the event behind the producer's segment, which was
    launched earlier)."""
    N = n_nodes
    pred: List[List[int]] = [[] for _ in range(N)]
    succ: List[List[int]] = [[] for _ in range(N)]
    for a, b in edges:
        pred[b].append(a)
        succ[a].append(b)
    # topological order; ties in creation (= capture issue) order
    deg = [len(p) for p in pred]
    heap = [i for i in range(N) if deg[i] == 0]
    heapq.heapify(heap)
    topo = []
    while heap:
        v = heapq.heappop(heap)
        topo.append(v)
        for w in succ[v]:
            deg[w] -= 1
            if deg[w] == 0:
                heapq.heappush(heap, w)
    if len(topo) != N:
        raise CtlError("SegmentReplay: the captured graph has a cycle")
    # chain cover: continue the chain of a predecessor that is still that chain's last node; else any chain whose last node is an
    # ancestor (stream order then adds nothing the edges do not already imply); else a new chain
    anc = [0] * N
    for v in topo:
        a = 0
        for p in pred[v]:
            a |= anc[p] | (1 << p)
        anc[v] = a
    chain_of: List[int] = [-1] * N
    tails: List[int] = []
    for v in topo:
        c = -1
        for p in pred[v]:
            if tails[chain_of[p]] == p:
                c = chain_of[p]
                break
        if c < 0:
            for cc, t in enumerate(tails):
                if (anc[v] >> t) & 1:
                    c = cc
                    break
        if c < 0:
            c = len(tails)
            tails.append(-1)
        chain_of[v] = c
        tails[c] = v
    K = len(tails)
    # segments: a node with a predecessor on another chain opens one (behind the wait), a node with a successor on another chain
    # closes one (the event is recorded right behind it: nothing later in the chain delays the other chain).  Segments are emitted when
    # they close, so an event's record is always issued before the waits on it
    open_seg: List[_Segment] = [None] * K
    segs: List[_Segment] = []
    seg_of: List[_Segment] = [None] * N

    def close(c):
        if open_seg[c] is not None and open_seg[c].nodes:
            segs.append(open_seg[c])
        open_seg[c] = None

    for v in topo:
        c = chain_of[v]
        cross = [p for p in pred[v] if chain_of[p] != c]
        if cross:
            close(c)
        if open_seg[c] is None:
            open_seg[c] = _Segment(c)
        s = open_seg[c]
        s.nodes.append(v)
        seg_of[v] = s
        for p in cross:
            if seg_of[p] not in s.waits:
                s.waits.append(seg_of[p])
        if any(chain_of[w] != c for w in succ[v]):
            close(c)
    for c in range(K):
        close(c)
    if emit == "open":                            # launch order = order of the segments' first nodes (also legal: a waited-for segment
        order = {v: i for i, v in enumerate(topo)}  # closed, hence opened, before its waiter opened)
        segs.sort(key=lambda s: order[s.nodes[0]])
    emitted = set()
    for s in segs:                                # (the invariant the replay loop relies on)
        if any(id(w) not in emitted for w in s.waits):
            raise CtlError("SegmentReplay: a segment waits for one that is launched later")
        emitted.add(id(s))
    return chain_of, segs


class SegmentReplay:
    """`SegmentReplay(g)` for a captured `torch.cuda.CUDAGraph(keep_graph=True)`; `.replay()` on the current stream is `g.replay()`.
    `g` must stay alive (it owns the captured graph's memory pool and the kernel-argument storage the clones were copied from)."""

    def __init__(self, cuda_graph: "torch.cuda.CUDAGraph", side_priority: int = 0, emit: str = "close", swap_chains: bool = False,
                 streams_of: "SegmentReplay" = None):
        hip = _lib()
        self._hip = hip
        self._owner = cuda_graph
        G = _VP(cuda_graph.raw_cuda_graph())
        n = C.c_size_t(0)
        _ck(hip.hipGraphGetNodes(G, None, C.byref(n)), "hipGraphGetNodes")
        arr = (_VP * n.value)()
        _ck(hip.hipGraphGetNodes(G, arr, C.byref(n)), "hipGraphGetNodes")
        nodes = [int(v) for v in arr]
        ne = C.c_size_t(0)
        _ck(hip.hipGraphGetEdges(G, None, None, C.byref(ne)), "hipGraphGetEdges")
        fr, to = (_VP * ne.value)(), (_VP * ne.value)()
        if ne.value:
            _ck(hip.hipGraphGetEdges(G, fr, to, C.byref(ne)), "hipGraphGetEdges")
        N = len(nodes)
        if N == 0:
            raise CtlError("SegmentReplay: the captured graph is empty")
        idx = {v: i for i, v in enumerate(nodes)}
        chain_of, segs = plan_segments(N, [(idx[int(a)], idx[int(b)]) for a, b in zip(fr, to)], emit=emit)
        K = max(chain_of) + 1
        if swap_chains and K == 2:                # (tuning aid: which chain runs on the stream the replay is launched on)
            for sg in segs:
                sg.chain = 1 - sg.chain
        if K > MAX_CHAINS:
            raise CtlError(f"SegmentReplay: the captured graph needs {K} chains (at most {MAX_CHAINS} are replayed)")
        self.segments, self.n_nodes, self.n_edges, self.n_chains = segs, N, int(ne.value), K
        self._events: List[_VP] = []
        try:
            for s in segs:
                clone = _VP()
                _ck(hip.hipGraphClone(C.byref(clone), G), "hipGraphClone")
                s.graph = clone
                keep = set(s.nodes)
                for v in range(N):
                    if v not in keep:
                        cn = _VP()
                        _ck(hip.hipGraphNodeFindInClone(C.byref(cn), _VP(nodes[v]), clone), "hipGraphNodeFindInClone")
                        _ck(hip.hipGraphDestroyNode(cn), "hipGraphDestroyNode")
                exe = _VP()
                _ck(hip.hipGraphInstantiate(C.byref(exe), clone, None, None, 0), "hipGraphInstantiate")
                s.exe = exe
            for s in segs:
                for w in s.waits:
                    if w.event is None:
                        w.event = self._new_event()
            self._fork = self._new_event()
            self._joins = [self._new_event() for _ in range(K - 1)]
        except Exception:
            self.close()
            raise
        self._side_priority = side_priority
        if streams_of is not None and len(streams_of._side) >= K - 1:      # (the graphs of one step object share the probed streams)
            self._pool = streams_of._pool
        else:
            self._pool = {"side": [torch.cuda.Stream(priority=side_priority) for _ in range(K - 1)], "checked": set(), "overlap": None}
        self.n_events = sum(1 for s in segs if s.event is not None)

    @property
    def _side(self):
        return self._pool["side"]

    @property
    def overlap(self):
        return self._pool["overlap"]

    def _new_event(self):
        ev = _VP()
        _ck(self._hip.hipEventCreateWithFlags(C.byref(ev), _EVENT_DISABLE_TIMING), "hipEventCreateWithFlags")
        self._events.append(ev)
        return ev

    def _ensure_chains_overlap(self, cur: "torch.cuda.Stream"):
        """Before the first replay on a launch stream: every other chain's stream must sit on another hardware queue than the launch
        stream (probe with idle kernels; on a collision take the next stream of torch's pool).  A chain that shares the launch stream's
        queue overlaps nothing: the fp32 step replays in 17.9 instead of 15.2 ms (tools/segments_probe.py)."""
        # (one probe per launch stream EVER: a caller alternating between launch streams must not pay the probe's device-wide synchronise on
        #  every replay -- ADVICE r4; `prepare()` runs it ahead of a timed region.  The probed side streams are kept PER launch stream
        #  -- replacing a shared side stream for launch stream B could put it on the queue of the already probed stream A)
        key = cur.cuda_stream
        if key in self._pool["checked"] or not self._side or torch.cuda.is_current_stream_capturing():
            return
        self._pool["checked"].add(key)
        side = list(self._side)
        report = []
        for i in range(len(side)):
            ratio, attempts = streams_overlap_ratio(cur, side[i]), 1
            while ratio > 1.5 and attempts < 8:
                side[i] = torch.cuda.Stream(priority=self._side_priority)
                ratio, attempts = streams_overlap_ratio(cur, side[i]), attempts + 1
            report.append({"probe_ratio": round(ratio, 2), "streams_tried": attempts, "overlap": ratio <= 1.5})
        self._pool.setdefault("side_of", {})[key] = side
        self._pool.setdefault("overlap_of", {})[key] = report
        self._pool["overlap"] = report

    def prepare(self, stream: "torch.cuda.Stream" = None):
        """Run the hardware-queue probe for `stream` (default: the current one) now, outside any timed region (it synchronises the device)."""
        self._ensure_chains_overlap(stream if stream is not None else torch.cuda.current_stream())

    def replay(self):
        hip = self._hip
        cur_stream = torch.cuda.current_stream()
        self._ensure_chains_overlap(cur_stream)
        cur = _VP(cur_stream.cuda_stream)
        side = self._pool.get("side_of", {}).get(cur_stream.cuda_stream, self._side)      # the streams probed against THIS launch stream
        raw = [cur] + [_VP(s.cuda_stream) for s in side[:self.n_chains - 1]]
        if len(raw) > 1:
            _ck(hip.hipEventRecord(self._fork, cur), "hipEventRecord")
            for st in raw[1:]:
                _ck(hip.hipStreamWaitEvent(st, self._fork, 0), "hipStreamWaitEvent")
        for s in self.segments:
            st = raw[s.chain]
            for w in s.waits:
                _ck(hip.hipStreamWaitEvent(st, w.event, 0), "hipStreamWaitEvent")
            _ck(hip.hipGraphLaunch(s.exe, st), "hipGraphLaunch")
            if s.event is not None:
                _ck(hip.hipEventRecord(s.event, st), "hipEventRecord")
        for st, ev in zip(raw[1:], self._joins):
            _ck(hip.hipEventRecord(ev, st), "hipEventRecord")
            _ck(hip.hipStreamWaitEvent(cur, ev, 0), "hipStreamWaitEvent")

    def describe(self) -> dict:
        return {"nodes": self.n_nodes, "edges": self.n_edges, "chains": self.n_chains, "segments": len(self.segments), "events": self.n_events,
                "largest_segment": max(len(s.nodes) for s in self.segments), "queue_probe": self.overlap}

    def close(self):
        hip = self._hip
        if any(s.exe is not None for s in getattr(self, "segments", [])) and torch.cuda.is_initialized():
            torch.cuda.synchronize()              # (nothing of these graphs may still be in flight)
        for s in getattr(self, "segments", []):
            if s.exe is not None:
                hip.hipGraphExecDestroy(s.exe)
                s.exe = None
            if s.graph is not None:
                hip.hipGraphDestroy(s.graph)
                s.graph = None
        for ev in getattr(self, "_events", []):
            hip.hipEventDestroy(ev)
        self._events = []

    def __del__(self):
        # not at interpreter shutdown: the HIP runtime may already be tearing down, and the process' exit frees everything anyway
        try:
            if sys is None or sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass
