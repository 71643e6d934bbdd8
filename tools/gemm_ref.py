"""Reference points for the projection GEMM shapes of a base frame on the GPU box: hipBLASLt bf16 and
fp32 (torch.nn.functional.linear) and a plain device copy of the fp32 operand bytes (the HBM floor a
fp32-in / fp32-out kernel cannot beat), next to this package's split-bf16 kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bevformer_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = "cuda:0"
shapes = [("sca_value_proj", 184950, 256, 1536), ("tsa_value_proj", 80000, 256, 1536), ("tsa_offs_attn", 40000, 512, 192),
          ("tsa_output_proj", 40000, 256, 256), ("sca_offs_attn", 40000, 256, 768), ("sca_output_proj", 40000, 256, 256),
          ("ffn_fc1", 40000, 256, 512), ("ffn_fc2", 40000, 512, 256)]
g = torch.Generator(device=DEV).manual_seed(0)
print(f"{'shape':18s} {'M':>7s} {'K':>4s} {'N':>5s} | {'ours split':>10s} {'ours bf16':>10s} {'blt fp32':>9s} {'blt bf16':>9s} {'copy':>7s}  (us)")
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) * 0.05
    b = torch.randn(N, device=DEV, generator=g)
    xb, wb, bb = x.bfloat16(), w.bfloat16(), b.bfloat16()
    res = {}
    with torch.no_grad():
        for mode in ("split", "bf16"):
            ops.set_gemm_mode(mode)
            res[mode] = timeit(lambda: ops.linear(x, w, b), 20)[0] * 1e6
        ops.set_gemm_mode("split")
        res["f32"] = timeit(lambda: torch.nn.functional.linear(x, w, b), 20)[0] * 1e6
        res["b16"] = timeit(lambda: torch.nn.functional.linear(xb, wb, bb), 20)[0] * 1e6
        y = torch.empty(M, N, device=DEV)
        src = torch.empty((M * K + M * N) // 2, device=DEV)
        dst = torch.empty_like(src)
        res["copy"] = timeit(lambda: dst.copy_(src), 20)[0] * 1e6
    print(f"{name:18s} {M:7d} {K:4d} {N:5d} | {res['split']:10.1f} {res['bf16']:10.1f} {res['f32']:9.1f} {res['b16']:9.1f} {res['copy']:7.1f}")
