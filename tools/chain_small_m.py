"""The row-chain kernels (csrc/linear_chain.h) and the launch sequences they replace at the row counts of a BEV tile
(Q / G rows per rank) and at the full grid: both workgroup shapes and the library's choice, timed inside a HIP graph.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bevformer_amd import ops  # noqa: E402
from gemm_small_m import timeit  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
w0, b0 = torch.randn(256, 256, device=DEV, generator=g) / 16, torch.randn(256, device=DEV, generator=g) * 0.1
fc1, fc2 = torch.nn.Linear(256, 512).to(DEV), torch.nn.Linear(512, 256).to(DEV)
wm, bm = torch.randn(768, 256, device=DEV, generator=g) / 16, torch.randn(768, device=DEV, generator=g) * 0.1
n0, n1 = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
print(f"{'rows':>6s} | FFN tail: 3 launches | chain 64-row | 32-row | library default || attention seam: 2 launches | 64-row | 32-row | library default   (us)")
with torch.no_grad():
    for M in (2500, 5000, 10000, 20000, 22500, 32768, 40000, 65536):
        rows, res = torch.randn(M, 256, device=DEV, generator=g), torch.randn(M, 256, device=DEV, generator=g)

        def three():
            x = ops.linear_layernorm(rows, w0, b0, res, n0)
            h = ops.linear(x, fc1.weight, fc1.bias, relu=True)
            return ops.linear_layernorm(h, fc2.weight, fc2.bias, x, n1)

        def two():
            x = ops.linear_layernorm(rows, w0, b0, res, n0)
            return ops.linear(x, wm, bm)
        cols = [timeit(three, 10)[0]]
        for shape in (1, 2, 0):
            with ops.using(chain_shape=shape):
                cols.append(timeit(lambda: ops.proj_ffn_chain(rows, w0, b0, res, n0, fc1, fc2, n1), 10)[0])
        cols.append(timeit(two, 10)[0])
        for shape in (1, 2, 0):
            with ops.using(chain_shape=shape):
                cols.append(timeit(lambda: ops.proj_ln_proj_chain(rows, w0, b0, res, n0, wm, bm), 10)[0])
        print(f"{M:6d} | " + " ".join(f"{c:8.1f}" for c in cols[:4]) + " || " + " ".join(f"{c:8.1f}" for c in cols[4:]))
