"""TemporalSelfAttention's sampling launch at the base grid (200 x 200 cells, 8 heads, 2 queue entries x 4 points, reference
points = cell centres + an ego-motion shift, offsets of the encoder's bias-grid size): the default kernel (tap lines from the
vector L1) against the LDS-tile kernel (fused_spec = 5), HIP events around ITER launches, interleaved rounds.  Library variants
(tools/build_variant.sh: tile height, diagnostic builds) are selected with BEVMSDA_LIBRARY as for every other tool."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevformer_amd import ops

DEV = torch.device("cuda:0")
ITER = 50


def main():
    gh = gw = int(os.environ.get("GRID", 200))
    spread = float(os.environ.get("SPREAD", 1.0))
    Q = gh * gw
    M, L, P, D, K = 8, 1, 4, 32, 2
    g = torch.Generator().manual_seed(7)
    shapes = torch.tensor([[gh, gw]])
    start = torch.zeros(1, dtype=torch.long)
    value = torch.randn(2, Q, M, D, generator=g)
    n_off = M * K * L * P * 2
    proj = torch.randn(Q, n_off + M * K * L * P, generator=g)
    proj[:, :n_off] *= 1.5 * spread
    ys, xs = torch.meshgrid((torch.arange(gh) + 0.5) / gh, (torch.arange(gw) + 0.5) / gw, indexing="ij")
    cur = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1)
    ref = torch.stack([cur + torch.tensor([0.013, -0.021]), cur], 1).reshape(Q, K, L, 2).contiguous()
    kw = dict(M=M, L=L, P=P, K=K, off_head=K * L * P * 2, off_k=L * P * 2, lg_head=K * L * P, lg_k=L * P,
              ref_mode=1, vmul=2, vadd=1, Q=Q)
    args = (value.to(DEV), shapes.to(DEV), start.to(DEV), proj.to(DEV), n_off, ref.to(DEV), None)

    def timed(spec):
        with ops.using(fused_spec=spec):
            ops.msda_fused(*args, grid_hw=(gh, gw), **kw)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(ITER):
                ops.msda_fused(*args, grid_hw=(gh, gw), **kw)
            ev[1].record()
            torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / ITER * 1e3

    with ops.using(fused_spec=5):
        a = ops.msda_fused(*args, grid_hw=(gh, gw), **kw)
    b = ops.msda_fused(*args, **kw)
    print("library", os.environ.get("BEVMSDA_LIBRARY", "default"), "grid", gh, "spread", spread, "bit-equal", bool(torch.equal(a, b)))
    for r in range(3):
        print(f"  round {r}: default {timed(0):.1f} us, LDS tiles {timed(5):.1f} us")


main()
