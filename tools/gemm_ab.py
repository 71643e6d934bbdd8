"""A/B of the projection kernels on the GPU box at the base-frame shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bevformer_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = "cuda:0"
shapes = [("sca_value_proj", 184950, 256, 0, 1536, 6), ("tsa_value_proj", 80000, 256, 0, 1536, 6),
          ("tsa_offs_attn(no pos)", 40000, 256, 256, 192, 1), ("tsa_output_proj", 40000, 256, 0, 256, 1),
          ("sca_offs_attn", 40000, 256, 0, 768, 1), ("ffn_fc1", 40000, 256, 0, 512, 1), ("ffn_fc2", 40000, 512, 0, 256, 1)]
g = torch.Generator(device=DEV).manual_seed(0)
for mode in ("split", "bf16"):
    ops.set_gemm_mode(mode)
    print(f"mode {mode}: {'shape':22s} first-kernel   dma-kernel   ws-kernel  pipe-kernel  areg-kernel (us, median of 20)   max |areg - first|")
    tot = [0.0, 0.0, 0.0, 0.0, 0.0]
    for name, M, K0, K1, N, G in shapes:
        x = torch.randn(M, K0, device=DEV, generator=g)
        x2 = torch.randn(M, K1, device=DEV, generator=g) if K1 else None
        w = torch.randn(N, K0 + K1, device=DEV, generator=g) * 0.05
        b = torch.randn(N, device=DEV, generator=g)
        t = []
        with torch.no_grad():
            outs = []
            for kern in ("first", "dma", "ws", "pipe", "areg"):
                ops.set_gemm_kernel(kern)
                outs.append(ops.linear(x, w, b, x2=x2, groups=G))
                t.append(timeit(lambda: ops.linear(x, w, b, x2=x2, groups=G), 20)[0] * 1e6)
        ops.set_gemm_kernel(None)
        for i in range(5):
            tot[i] += t[i] * (1 if G > 1 else 6)
        print(f"   {name:22s} {t[0]:10.1f} {t[1]:10.1f} {t[2]:10.1f} {t[3]:10.1f} {t[4]:10.1f}      {(outs[4] - outs[0]).abs().max().item():.3e}")
    print(f"   per frame (hoisted x1, per-layer x6): {tot[0]:.0f} vs {tot[1]:.0f} vs {tot[2]:.0f} vs {tot[3]:.0f} vs {tot[4]:.0f} us")
