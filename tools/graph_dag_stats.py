#!/usr/bin/env python3
"""Shape of the captured whole-step graph: nodes, edges, forks (out-degree > 1), joins (in-degree > 1), by node kind
(hipGraphDebugDotPrint of the captured graph, parsed).  Usage: graph_dag_stats.py [fp32|bf16] [--single-stream]"""
import sys, os, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

_Base = torch.cuda.CUDAGraph


class DebugGraph(_Base):
    def __new__(cls, *a, **k):
        return super().__new__(cls, keep_graph=True)

    def __init__(self, *a, **k):
        super().__init__(keep_graph=True)


torch.cuda.CUDAGraph = DebugGraph
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(0)
solver = AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard", image_ch=1, num_classes=4, learning_rate=1e-4, use_gpu=True,
                                               compute_dtype=dtype)
if "--single-stream" in sys.argv:
    solver.two_streams = False
IMG, SEG, _ = bench.MASKS["targeted" if dtype == "bf16" else "dropout"]
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, device)
for _ in range(3):
    solver.cooperative_step(clean, label, noisy, IMG, SEG)
g = CooperativeStepGraph(solver, IMG, SEG)
g(clean, label, noisy)
torch.cuda.synchronize()
e = next(iter(g.entries.values()))
import ctypes as C
hip = C.CDLL("libamdhip64.so")
graph = C.c_void_p(e.graph.raw_cuda_graph())
n = C.c_size_t(0)
assert hip.hipGraphGetNodes(graph, None, C.byref(n)) == 0
nodes_a = (C.c_void_p * n.value)()
assert hip.hipGraphGetNodes(graph, nodes_a, C.byref(n)) == 0
ne = C.c_size_t(0)
assert hip.hipGraphGetEdges(graph, None, None, C.byref(ne)) == 0
fr, to = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
assert hip.hipGraphGetEdges(graph, fr, to, C.byref(ne)) == 0
KIND = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}
kinds = {}
for v in nodes_a:
    t = C.c_int(-1)
    hip.hipGraphNodeGetType(C.c_void_p(v), C.byref(t))
    kinds[v] = KIND.get(t.value, str(t.value))
edges = list(zip(fr, to))
indeg, outdeg = collections.Counter(b for _, b in edges), collections.Counter(a for a, _ in edges)
nodes = list(nodes_a)
print(f"{dtype}{' one chain' if not solver.two_streams else ''}: {len(nodes)} nodes, {len(edges)} edges; forks (out-degree > 1): {sum(v > 1 for v in outdeg.values())}; "
      f"joins (in-degree > 1): {sum(v > 1 for v in indeg.values())}; roots {sum(indeg[v] == 0 for v in nodes)}, leaves {sum(outdeg[v] == 0 for v in nodes)}")
print("kinds of the join nodes:", collections.Counter(kinds[v] for v in nodes if indeg[v] > 1).most_common())
print("kinds of the fork nodes:", collections.Counter(kinds[v] for v in nodes if outdeg[v] > 1).most_common())
print("node kinds:", collections.Counter(kinds.values()).most_common())
print("in-degree histogram:", sorted(collections.Counter(indeg[v] for v in nodes).items()), " out-degree histogram:", sorted(collections.Counter(outdeg[v] for v in nodes).items()))
# width profile: longest-path level of every node, nodes per level
succ = collections.defaultdict(list)
for a_, b_ in edges: succ[a_].append(b_)
level = {}
order = [v for v in nodes if indeg[v] == 0]
deg = dict(indeg)
i = 0
for v in order: level[v] = 0
while i < len(order):
    v = order[i]; i += 1
    for w in succ[v]:
        level[w] = max(level.get(w, 0), level[v] + 1)
        deg[w] -= 1
        if deg[w] == 0: order.append(w)
per = collections.Counter(level.values())
print(f"longest path {max(level.values()) + 1} nodes; levels with 1 node: {sum(v == 1 for v in per.values())}, 2: {sum(v == 2 for v in per.values())}, 3+: {sum(v >= 3 for v in per.values())}")
