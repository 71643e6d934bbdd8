"""Projection-GEMM micro-benchmark on the GPU box: the eight nn.Linear shapes of one
bevformer_base encoder layer through ``ops.linear`` (hand-written MFMA kernel, modes
``split`` / ``bf16``) and through torch / hipBLASLt fp32 (``native``).  One JSON object per
line; also written to gpurun_out/gbench.json.

    python tools/gbench.py [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import ops  # noqa: E402

DEV = "cuda:0"
# (tag, rows, K0, K1, N, relu, addend on the second source)
SHAPES = [
    ("sca_value_proj", 6 * 30825, 256, 0, 256, False, False),
    ("tsa_value_proj", 2 * 40000, 256, 0, 256, False, False),
    ("tsa_offs_attn", 40000, 256, 256, 192, False, True),
    ("sca_offs_attn", 40000, 256, 0, 768, False, False),
    ("output_proj", 40000, 256, 0, 256, False, False),
    ("ffn_fc1", 40000, 256, 0, 512, True, False),
    ("ffn_fc2", 40000, 512, 0, 256, False, False),
]


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="comma-separated op tags")
    ap.add_argument("--modes", default="native,split,bf16")
    ap.add_argument("--variants", default="", help="comma-separated launch variants of the MFMA kernel "
                    "(include/bevmsda.h); empty = library default")
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    results = []
    for tag, M, K0, K1, N, relu, add in SHAPES:
        if args.only and tag not in args.only.split(","):
            continue
        x0 = torch.randn(M, K0, generator=g).to(DEV)
        x1 = torch.randn(M, K1, generator=g).to(DEV) if K1 else None
        a1 = torch.randn(M, K1, generator=g).to(DEV) if add else None
        w = (torch.randn(N, K0 + K1, generator=g) * 0.05).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        flops = 2.0 * M * N * (K0 + K1)
        nbytes = 4.0 * (M * (K0 + K1) + (M * K1 if add else 0) + N * (K0 + K1) + M * N)
        state = {"want": None}

        def run(mode):
            with torch.no_grad():
                if mode == "native":
                    xa = x0 if x1 is None else torch.cat([x0, x1 + a1 if add else x1], -1)
                    y = torch.nn.functional.linear(xa, w, b)
                    return torch.relu_(y) if relu else y
                return ops.linear(x0, w, b, relu=relu, x2=x1, x2_add=a1)

        for mode in ("native", "split", "bf16"):
            if mode != "native" and mode not in args.modes.split(","):
                continue
            ops.set_gemm_mode(mode)
            variants = [None] if mode == "native" or not args.variants \
                else [int(v) for v in args.variants.split(",")]
            for variant in variants:
                ops.set_gemm_variant(variant)
                y = run(mode)
                if mode == "native":
                    state["want"] = y
                    err = 0.0
                else:
                    err = ((y - state["want"]).abs().max() / state["want"].abs().max()).item()
                if mode == "native" and "native" not in args.modes.split(","):
                    continue
                med, mn = timeit(lambda: run(mode), args.iters)
                r = dict(op=tag, mode=mode, variant=variant, M=M, N=N, K=K0 + K1, us=med * 1e6,
                         min_us=mn * 1e6, TFLOPs=flops / med / 1e12, alg_GBs=nbytes / med / 1e9,
                         alg_MB=nbytes / 1e6, max_err_vs_native=err)
                print(json.dumps(r), flush=True)
                results.append(r)
        ops.set_gemm_variant(None, pack=True)
        del x0, x1, a1, w, b
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "gbench.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
