"""How fast are the library's TN weight-gradient GEMMs (grad_W = G^T X, reduction over the rows) at the
encoder's shapes?  fp32 (what the autograd path runs today) vs bf16 operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit  # noqa: E402

DEV = "cuda:0"
shapes = [("sca_value_proj", 184950, 256, 256), ("tsa_value_proj", 80000, 256, 256), ("tsa_offs_attn", 40000, 192, 512),
          ("sca_offs_attn", 40000, 768, 256), ("output_proj", 40000, 256, 256), ("ffn_fc1", 40000, 512, 256),
          ("ffn_fc2", 40000, 256, 512)]
g_ = torch.Generator(device=DEV).manual_seed(0)
print(f"{'shape':16s} {'M':>7s} {'N':>4s} {'K':>4s} | fp32 TN   bf16 TN   bf16 cast (us)")
for name, M, N, K in shapes:
    g = torch.randn(M, N, device=DEV, generator=g_)
    x = torch.randn(M, K, device=DEV, generator=g_)
    gb, xb = g.bfloat16(), x.bfloat16()
    t32 = timeit(lambda: g.t() @ x, 10)[0] * 1e6
    t16 = timeit(lambda: gb.t() @ xb, 10)[0] * 1e6
    tc = timeit(lambda: (g.bfloat16(), x.bfloat16()), 10)[0] * 1e6
    print(f"{name:16s} {M:7d} {N:4d} {K:4d} | {t32:8.1f} {t16:8.1f} {tc:8.1f}")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import ops  # noqa: E402
print("this package's kernel (csrc/wgrad_mfma.h), weight + bias gradient, incl. the zero fills:")
for name, M, N, K in shapes:
    g = torch.randn(M, N, device=DEV, generator=g_)
    x = torch.randn(M, K, device=DEV, generator=g_)
    t = timeit(lambda: ops.linear_wgrad(g, x, True), 10)[0] * 1e6
    print(f"{name:16s} {M:7d} {N:4d} {K:4d} | {t:8.1f}")
