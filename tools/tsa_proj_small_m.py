"""TemporalSelfAttention's merged [sampling_offsets ; attention_weights] projection (K = 512 from two sources, addend on the second,
N = 192) at tile-sized and full row counts: first kernel vs row-panel kernel, timed inside a HIP graph.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bevformer_amd import ops  # noqa: E402
from gemm_small_m import timeit  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
w, b = torch.randn(192, 512, device=DEV, generator=g) / 16, torch.randn(192, device=DEV, generator=g) * 0.1
print(f"{'rows':>6s} | first kernel | panel64 | panel128 | first64 (64 x 256 tiles) (us)   max |panel64 - first|  max |first64 - first|")
with torch.no_grad():
    for M in (2500, 5000, 10000, 20000, 40000):
        first, q, pos = (torch.randn(M, 256, device=DEV, generator=g) for _ in range(3))
        cols, outs = [], []
        for kern in ("first", "panel64", "panel128", "first64"):
            with ops.using(gemm_kernel=kern):
                f = lambda: ops.linear(first, w, b, x2=q, x2_add=pos)
                outs.append(f())
                cols.append(timeit(f, 10)[0])
        print(f"{M:6d} | {cols[0]:8.1f} {cols[1]:8.1f} {cols[2]:8.1f} {cols[3]:8.1f}    {(outs[1] - outs[0]).abs().max().item():.2e}   {(outs[3] - outs[0]).abs().max().item():.2e}")
