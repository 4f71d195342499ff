#!/usr/bin/env python3
"""What does an edge between the two launch chains cost inside a hipGraph replay, against the same event wait between two streams?
Two chains of `n` small kernels each (elementwise passes over 4 MB, ~5 us); every `n / k` kernels each chain waits for the other's progress."""
import sys, os, time
import torch

dev = torch.device("cuda", 0)
N = 40
a = [torch.randn(1 << 20, device=dev) for _ in range(2)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def body(k, s0, s1):
    """issue on s0 / s1 (s0 is the origin: s1 forks from it and joins back at the end)"""
    s1.wait_stream(s0)
    every = (N // k) if k else 0
    for i in range(N):
        with torch.cuda.stream(s0):
            a[0].mul_(1.0001)
        with torch.cuda.stream(s1):
            a[1].mul_(1.0001)
        if every and (i + 1) % every == 0 and i + 1 < N:
            e0, e1 = torch.cuda.Event(), torch.cuda.Event()
            e0.record(s0); e1.record(s1)
            s1.wait_event(e0); s0.wait_event(e1)
    s0.wait_stream(s1)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / reps


for k in (0, 1, 4, 10, 20, 40):
    def eager():
        sa.wait_stream(torch.cuda.current_stream())
        body(k, sa, sb)
        torch.cuda.current_stream().wait_stream(sa)
    te = timed(eager)
    g = torch.cuda.CUDAGraph()
    sa.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=sa):
        for _ in range(5):
            body(k, sa, sb)
    torch.cuda.current_stream().wait_stream(sa)
    tg = timed(g.replay) / 5
    # one chain for scale
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=sa):
        for _ in range(5):
            for i in range(N):
                a[0].mul_(1.0001); a[1].mul_(1.0001)
    t1 = timed(g1.replay) / 5
    print(f"2 x {N} kernels, {k:2d} cross waits each way: eager {te:7.1f} us (host-bound?)  graph {tg:7.1f} us   one-chain graph {t1:7.1f} us   "
          f"-> per cross pair in the graph {(tg - t1 / 2) / max(k, 1):6.2f} us over the ideal {t1 / 2:.1f}", flush=True)

# ---- the same work as LINEAR graphs (one per chain segment) replayed on two streams with event waits in between
print("linear segment graphs on two streams:")
for k in (0, 1, 4, 10, 20, 40):
    nseg = k if k else 1
    per = N // nseg
    segs = []
    for ch in (0, 1):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sa):
            for i in range(per):
                a[ch].mul_(1.0001)
        segs.append(g)
    evs = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(nseg)]

    def run():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        for i in range(nseg):
            with torch.cuda.stream(sa):
                segs[0].replay()
            with torch.cuda.stream(sb):
                segs[1].replay()
            if k and i + 1 < nseg:
                e0, e1 = evs[i]
                e0.record(sa); e1.record(sb)
                sb.wait_event(e0); sa.wait_event(e1)
        cur.wait_stream(sa); cur.wait_stream(sb)
    t = timed(run)
    print(f"2 x {N} kernels in {nseg} segments per chain, {max(nseg - 1, 0)} cross waits each way: {t:7.1f} us", flush=True)
