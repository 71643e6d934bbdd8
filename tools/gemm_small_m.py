"""Projection kernels at the row counts of a BEV tile (Q / G rows per rank): first kernel vs the
software-pipelined one.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bevformer_amd import ops  # noqa: E402


def timeit(fn, iters):
    """us per call of ``fn`` from the replay of a HIP graph holding ``iters`` back-to-back calls (no host
    launch latency between the kernels: at these sizes an eager loop measures the host)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
print(f"{'M':>7s} {'K':>4s} {'N':>5s} | first   pipe (us per call inside a 30-call HIP graph)")
for M in (2500, 5000, 10000, 20000, 40000):
    for K, N in ((256, 256), (256, 512), (512, 256), (256, 768)):
        x = torch.randn(M, K, device=DEV, generator=g)
        w = torch.randn(N, K, device=DEV, generator=g) * 0.05
        b = torch.randn(N, device=DEV, generator=g)
        t = []
        with torch.no_grad():
            for kern in ("first", "pipe"):
                ops.set_gemm_kernel(kern)
                t.append(timeit(lambda: ops.linear(x, w, b), 30)[0])
        ops.set_gemm_kernel(None)
        print(f"{M:7d} {K:4d} {N:5d} | {t[0]:6.1f} {t[1]:6.1f}")
