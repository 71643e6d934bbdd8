"""A/B of the seam between two encoder layers at the base row count (40,000 rows, gather form): the chain kernel alone, the
chain kernel with the next layer's TemporalSelfAttention projection behind it (csrc/linear_chain.h TP), and the stand-alone
two-source projection it replaces — HIP events around ITER launches each, interleaved rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevformer_amd import ops

DEV = torch.device("cuda:0")
M = int(os.environ.get("M", 40000))
ITER = 50


def rand(*s, seed):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed)).to(DEV)


def main():
    mode = os.environ.get("GEMM", "split")
    ops.set_gemm_mode(mode)
    R = 45960
    rows = rand(R, 256, seed=1)
    g = torch.Generator().manual_seed(2)
    idx = torch.randint(0, R, (M, 2), generator=g, dtype=torch.int32)
    idx[torch.rand(M, generator=g) < 0.6, 1] = -1
    scale = (1.0 / (idx >= 0).sum(1).clamp(min=1).float()).to(DEV)
    idx = idx.to(DEV)
    w0, b0, res = rand(256, 256, seed=3) / 16, rand(256, seed=4) * 0.1, rand(M, 256, seed=5)
    fc1, fc2 = torch.nn.Linear(256, 512).to(DEV), torch.nn.Linear(512, 256).to(DEV)
    n0, n1 = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    first, pos = rand(1, M, 256, seed=6), rand(1, M, 256, seed=7)
    w3, b3 = rand(192, 512, seed=8) / 16, rand(192, seed=9) * 0.1
    shape = int(os.environ.get("SHAPE", 0))

    def plain():
        return ops.proj_ffn_chain(rows, w0, b0, res, n0, fc1, fc2, n1, gather=(idx, scale))

    def tail():
        return ops.proj_ffn_chain(rows, w0, b0, res, n0, fc1, fc2, n1, gather=(idx, scale), tail=(first, pos, w3, b3))

    y = None

    def alone():
        return ops.linear(first, w3, b3, x2=y, x2_add=pos)

    def timed(fn):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(ITER):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / ITER * 1e3

    with torch.no_grad(), ops.using(ln_fuse=True, chain_shape=shape):
        y = plain().view(1, M, 256)
        for f in (plain, tail, alone):
            f()
        for r in range(4):
            print(f"{mode} M={M} shape={shape} round {r}: chain {timed(plain):.1f} us, chain+tail {timed(tail):.1f} us, "
                  f"stand-alone projection {timed(alone):.1f} us")


main()
