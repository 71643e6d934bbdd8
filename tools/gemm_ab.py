"""A/B of the projection kernels on the GPU box at the base-frame shapes: the first kernel (linear_mfma.h)
against the row-panel kernel (linear_panel.h) in both panel shapes, interleaved rounds in one process,
median and minimum of the per-round times; then the projection + add + LayerNorm sequences of a layer as two
launches against the fused epilogue.

    python tools/gemm_ab.py [--rounds 5] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bevformer_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = "cuda:0"
KERNELS = ("first", "panel64w2", "panel64w6", "panel128")
SHAPES = [("sca_value_proj", 184950, 256, 0, 1536, 6, False), ("tsa_value_proj", 80000, 256, 0, 1536, 6, False),
          ("tsa_offs_attn", 40000, 256, 256, 192, 1, False), ("tsa_output_proj", 40000, 256, 0, 256, 1, False),
          ("sca_offs_attn", 40000, 256, 0, 768, 1, False), ("ffn_fc1", 40000, 256, 0, 512, 1, True),
          ("ffn_fc2", 40000, 512, 0, 256, 1, False), ("tile_5000x256x256", 5000, 256, 0, 256, 1, False)]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--modes", default="split,bf16")
    args = ap.parse_args()
    g = torch.Generator(device=DEV).manual_seed(0)
    for mode in args.modes.split(","):
        ops.set_gemm_mode(mode)
        print(f"mode {mode}: {'shape':22s}" + "".join(f"{k:>20s}" for k in KERNELS) + "   (us: median / min over rounds)   max |panel - first|")
        tot = {k: 0.0 for k in KERNELS}
        for name, M, K0, K1, N, G, relu in SHAPES:
            x = torch.randn(M, K0, device=DEV, generator=g)
            x2 = torch.randn(M, K1, device=DEV, generator=g) if K1 else None
            pos = torch.randn(M, K1, device=DEV, generator=g) if K1 else None
            w = torch.randn(N, K0 + K1, device=DEV, generator=g) * 0.05
            b = torch.randn(N, device=DEV, generator=g)
            ts = {k: [] for k in KERNELS}
            outs = {}
            with torch.no_grad():
                for r in range(args.rounds):
                    for k in KERNELS:
                        ops.set_gemm_kernel(k)
                        fn = lambda: ops.linear(x, w, b, relu=relu, x2=x2, x2_add=pos, groups=G)
                        if r == 0:
                            outs[k] = fn()
                        ts[k].append(timeit(fn, args.iters)[0] * 1e6)
            ops.set_gemm_kernel(None)
            for k in KERNELS:
                tot[k] += med(ts[k]) * (1 if G > 1 or name.startswith("tile") else 6) * (0 if name.startswith("tile") else 1)
            diff = max((outs[k] - outs["first"]).abs().max().item() for k in KERNELS[1:])
            print(f"   {name:22s}" + "".join(f"{med(ts[k]):11.1f} /{min(ts[k]):7.1f}" for k in KERNELS) + f"      {diff:.3e}")
        print("   per frame (hoisted x1, per-layer x6): " + " vs ".join(f"{tot[k]:.0f}" for k in KERNELS) + " us")

    # projection + residual + LayerNorm: two launches against the fused epilogue (split mode)
    ops.set_gemm_mode("split")
    norm = torch.nn.LayerNorm(256).to(DEV)
    print("projection + add + LayerNorm (split): two launches (first kernel) | two launches (panel) | fused epilogue panel64 | panel128   (us)")
    for name, M, K in (("output_proj+LN", 40000, 256), ("ffn_fc2+LN", 40000, 512), ("tile 5000 rows", 5000, 256)):
        x = torch.randn(M, K, device=DEV, generator=g)
        res = torch.randn(M, 256, device=DEV, generator=g)
        w = torch.randn(256, K, device=DEV, generator=g) * 0.05
        b = torch.randn(256, device=DEV, generator=g)
        cols = []
        with torch.no_grad():
            for k in ("first", "panel64"):
                ops.set_gemm_kernel(k)
                cols.append(med([timeit(lambda: ops.add_layernorm(ops.linear(x, w, b), res, norm.weight, norm.bias, norm.eps),
                                        args.iters)[0] * 1e6 for _ in range(args.rounds)]))
            for k in ("panel64", "panel128"):
                ops.set_gemm_kernel(k)
                cols.append(med([timeit(lambda: ops.linear_layernorm(x, w, b, res, norm), args.iters)[0] * 1e6
                                 for _ in range(args.rounds)]))
            ops.set_gemm_kernel(None)
        print(f"   {name:22s}" + "".join(f"{c:12.1f}" for c in cols))


if __name__ == "__main__":
    main()
