"""The SCA sampling launch the bench times (base frame: ~46 k ragged rows, shared projection rows, device-side row count) by
itself: HIP events around ITER launches.  BEVMSDA_LIBRARY selects a library variant (tools/build_variant.sh), e.g. the
diagnostic builds of csrc/msda_d32.h (-DBEVMSDA_SCA_DIAG=1: no softmax arithmetic, 2: no front-end loads either)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevformer_amd import ops
from bevformer_amd import synthetic as S
from bevformer_amd.modules import geometry as G

DEV = torch.device("cuda:0")
ITER = 30


def main():
    name = "base"
    w = S.WORKLOADS[name]
    Q = w["bev_h"] * w["bev_w"]
    M, L, P, D = 8, 4, 8, 32
    g = torch.Generator().manual_seed(0)
    shapes, start = S.level_tensors(name)
    Sv = int(shapes.prod(1).sum())
    value = torch.randn(S.NUM_CAMS, Sv, M, D, generator=g).to(DEV)
    proj = torch.randn(Q, M * L * P * 3, generator=g)
    n_off = M * L * P * 2
    proj[:, :n_off] *= 4.0
    proj = proj.to(DEV)
    pl = G.DevicePlanner(w["bev_h"], w["bev_w"], 1, S.PC_RANGE, 4, S.NUM_CAMS, DEV, row_order="image")
    plan = pl.plan(S.make_img_metas(name))
    kw = dict(M=M, L=L, P=P, K=1, off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0, vmul=1, vadd=0)
    sh, st = shapes.to(DEV), start.to(DEV)

    def run():
        return ops.msda_fused(value, sh, st, proj, n_off, plan.row_ref.reshape(-1, 1, 4, 2), plan.row_batch, row_src=plan.row_query32,
                              nrows=plan.nrows_dev, launch_rows=plan.launch_rows, **kw)
    run()
    for r in range(3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(ITER):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        print("library", os.path.basename(os.environ.get("BEVMSDA_LIBRARY", "default")), f"round {r}: {ev[0].elapsed_time(ev[1]) / ITER * 1e3:.1f} us",
              "rows", int(plan.nrows_dev.item()))


main()
