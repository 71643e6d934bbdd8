"""A/B of the projection epilogues on the GPU box (round 5): the row-panel kernel with its bias fragments loaded per
column tile (default) against the round-4 epilogue (bias loaded in front of every 16-byte store: 16 waited store round
trips per tile), the dripping-store and deeper-weight-prefetch variants — on the two hoisted value projections of a base
frame — and the first kernel's per-layer projections.  Interleaved rounds in one process, median / min per kernel.

    python tools/gemm_epilogue_ab.py [--rounds 5] [--iters 10]
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
# "" = the default build; e4 = the round-4 epilogue (bias loaded per piece); e2 = weight fragments 4 steps ahead;
# sN = phase skew of the column sweep, N x 1024 clocks (linear_panel.h); p = persistent grid
VARIANTS = ("", "e4")


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
    shapes = [("sca_value_proj", 184950, 1536, 6), ("tsa_value_proj", 80000, 1536, 6), ("sca_value_proj_small4", 117360, 768, 3)]
    for mode in args.modes.split(","):
        ops.set_gemm_mode(mode)
        for name, M, N, G in shapes:
            x = torch.randn(M, 256, device=DEV, generator=g)
            w = torch.randn(N, 256, device=DEV, generator=g) * 0.05
            b = torch.randn(N, device=DEV, generator=g)
            for store in (torch.float32, torch.bfloat16):
                kernels = [f"panel{bm}{sfx}" for bm in (128, 64) for sfx in VARIANTS] + ["panel128d2", "panel128d4"]
                ts = {k: [] for k in kernels}
                outs = {}
                with torch.no_grad():
                    for r in range(args.rounds):
                        for k in kernels:
                            ops.set_gemm_kernel(k)
                            fn = lambda: ops.linear(x, w, b, groups=G, out_dtype=store)   # noqa: E731
                            if r == 0:
                                outs[k] = fn()
                                assert outs[k] is not None, k
                            ts[k].append(timeit(fn, args.iters)[0] * 1e6)
                ops.set_gemm_kernel(None)
                same = all(torch.equal(outs[k], outs[kernels[0]]) for k in kernels if k.startswith("panel128")) and \
                    all(torch.equal(outs[k], outs["panel64"]) for k in kernels if k.startswith("panel64"))
                gb = (M * 256 * 4 + M * N * (4 if store == torch.float32 else 2)) / 1e9
                print(f"{mode:5s} {name:22s} out {str(store)[6:]:8s} bit-equal across epilogues: {same}")
                for k in kernels:
                    print(f"      {k:12s} {med(ts[k]):8.1f} / {min(ts[k]):8.1f} us   {gb / (med(ts[k]) * 1e-6) / 1e3:5.2f} TB/s   "
                          f"{2.0 * M * N * 256 / (med(ts[k]) * 1e-6) / 1e12:6.1f} TFLOP/s")
    # the first kernel's plain projections with a bias (its epilogue got the same treatment)
    ops.set_gemm_mode("split")
    ops.set_gemm_kernel("first")
    for name, M, K0, K1, N in (("tsa_offs_attn", 40000, 256, 256, 192), ("tsa_output_proj", 40000, 256, 0, 256),
                               ("sca_offs_attn", 40000, 256, 0, 768), ("ffn_fc1", 40000, 256, 0, 512)):
        x = torch.randn(M, K0, device=DEV, generator=g)
        x2 = torch.randn(M, K1, device=DEV, generator=g) if K1 else None
        pos = torch.randn(M, K1, device=DEV, generator=g) if K1 else None
        w = torch.randn(N, K0 + K1, device=DEV, generator=g) * 0.05
        b = torch.randn(N, device=DEV, generator=g)
        with torch.no_grad():
            t_b = med([timeit(lambda: ops.linear(x, w, b, x2=x2, x2_add=pos), args.iters)[0] * 1e6 for _ in range(args.rounds)])
            t_n = med([timeit(lambda: ops.linear(x, w, None, x2=x2, x2_add=pos), args.iters)[0] * 1e6 for _ in range(args.rounds)])
            ref = torch.nn.functional.linear(torch.cat([x, x2 + pos], -1) if K1 else x, w, b)
            err = (ops.linear(x, w, b, x2=x2, x2_add=pos) - ref).abs().max().item()
        print(f"first kernel {name:18s} with bias {t_b:7.1f} us   without {t_n:7.1f} us   max |y - torch| {err:.2e}")
    ops.set_gemm_kernel(None)


if __name__ == "__main__":
    main()
