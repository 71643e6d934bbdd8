"""Can the L1-bound sampling kernel and an MFMA / HBM-bound projection share the chip?  

The base SCA sampling launch (45,960 rows, fused kernel) on one stream, one layer's camera-value projection
(184,950 x 256 -> 256) on another; each alone, then both at once, with the sampling kernel's occupancy capped through
unused dynamic LDS (desc->reserved[4]) so that workgroups of the projection can become resident beside it.
A pair that co-schedules well finishes in ~max(t_a, t_b); one that serialises takes t_a + t_b.

    python tools/overlap_probe.py        (GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import ops  # noqa: E402
from bevformer_amd import synthetic as S  # noqa: E402
from bevformer_amd.modules import geometry as G  # noqa: E402

DEV = torch.device("cuda:0")


def wall(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    name = "base"
    w = S.WORKLOADS[name]
    Q = w["bev_h"] * w["bev_w"]
    M, L, P, D = 8, 4, 8, 32
    g = torch.Generator().manual_seed(3)
    shapes, start = S.level_tensors(name)
    Sv = int(shapes.prod(1).sum())
    value = torch.randn(S.NUM_CAMS, Sv, M, D, generator=g).to(DEV)
    n_off = M * L * P * 2
    pl = G.DevicePlanner(w["bev_h"], w["bev_w"], 1, S.PC_RANGE, 4, S.NUM_CAMS, DEV, row_order="image")
    host = pl.plan(S.make_img_metas(name)).materialize()
    proj = torch.randn(Q, M * L * P * 3, generator=g)
    proj[:, :n_off] *= 4.0
    proj = proj.to(DEV)
    kw = dict(M=M, L=L, P=P, K=1, off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0, vmul=1, vadd=0)
    sargs = (value, shapes.to(DEV), start.to(DEV), proj, n_off, host.row_ref.reshape(-1, 1, 4, 2), host.row_batch)
    feats = torch.randn(S.NUM_CAMS * Sv, 256, device=DEV)
    wv = torch.randn(256, 256, device=DEV) * 0.05
    bv = torch.randn(256, device=DEV)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def sample(pad):
        with ops.using(fused_lds_pad_kb=pad):
            return ops.msda_fused(*sargs, row_src=host.row_query32, **kw)

    def gemm(kern):
        with ops.using(gemm_kernel=kern):
            return ops.linear(feats, wv, bv)

    def both(pad, kern, first):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        order = ((s1, lambda: sample(pad)), (s2, lambda: gemm(kern)))
        for st, fn in (order if first == "sample" else order[::-1]):
            with torch.cuda.stream(st):
                fn()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    with torch.no_grad():
        print("sampling kernel alone (us), by LDS pad (0 = default occupancy, 54 KiB = 2 workgroups / CU, 64 = 2 (cap), ...):")
        ts = {}
        for pad in (0, 40, 54, 64):
            ts[pad] = wall(lambda: sample(pad))
            print(f"   pad {pad:2d} KiB: {ts[pad]:7.1f}")
        tg = {}
        for kern in ("first", "panel64"):
            tg[kern] = wall(lambda: gemm(kern))
            print(f"projection {kern:8s} alone: {tg[kern]:7.1f} us")
        print("both at once (two streams): wall us | sum of the two alone | max of the two alone")
        for kern in ("first", "panel64"):
            for pad in (0, 40, 54, 64):
                for first in ("sample", "gemm"):
                    t = wall(lambda: both(pad, kern, first))
                    print(f"   {kern:8s} pad {pad:2d} KiB, {first:6s} launched first: {t:7.1f} | {ts[pad] + tg[kern]:7.1f} | {max(ts[pad], tg[kern]):7.1f}")


if __name__ == "__main__":
    main()
