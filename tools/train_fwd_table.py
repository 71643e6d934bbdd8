"""Gradient-error table that settles which GEMM the autograd path's FORWARD projections use.

BASELINE configs[2] (small4: 150x150 BEV, 4 levels, 3 layers), forward + backward w.r.t. BEV queries,
camera features and every parameter.  Reference: the oracle evaluated in float64 on the CPU.  Rows: the
oracle in float32 (what "fp32 arithmetic elsewhere" buys), this package with (a) library fp32 forward
GEMMs, (b) split-bf16 MFMA forward GEMMs, (c) every GEMM on the library (mode native).  Columns: relative
L2 error of the output, of d/d(bev_query), d/d(feat), and the worst / median parameter gradient.

usage (GPU box): python tools/train_fwd_table.py [workload]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bevformer_amd import ops, synthetic as S  # noqa: E402
from oracle import bevformer_cpu as O  # noqa: E402
from helpers import build_pair  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "small4"
DEV = torch.device("cuda:0")
torch.set_num_threads(16)
enc, sd = build_pair(name, device=DEV)
q, f, kw = S.make_inputs(name, seed=0, temporal=True)
gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(5))


def oracle(dtype):
    leaves = {k: v.detach().to(dtype).requires_grad_(True) if v.is_floating_point() else v for k, v in sd.items()}
    qc, fc = q.detach().clone().to(dtype).requires_grad_(True), f.detach().clone().to(dtype).requires_grad_(True)
    kwc = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
    out = O.encoder_forward(leaves, qc, fc, pc_range=S.PC_RANGE, **kwc)
    out.backward(gout.to(dtype))
    g = {"bev_query": qc.grad, "feat": fc.grad}
    g.update({k: v.grad for k, v in leaves.items() if torch.is_tensor(v) and v.grad is not None})
    return out.detach(), g


def package():
    for p in enc.parameters():
        p.requires_grad_(True)
        p.grad = None
    qd, fd = q.detach().to(DEV).requires_grad_(True), f.detach().to(DEV).requires_grad_(True)
    kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    out = enc(qd, fd, fd, **kwd)
    out.backward(gout.to(DEV))
    g = {"bev_query": qd.grad.cpu(), "feat": fd.grad.cpu()}
    g.update({k: p.grad.cpu() for k, p in enc.named_parameters()})
    return out.detach().cpu(), g


def row(label, out, g, ref_out, ref_g):
    def l2(a, b):
        return ((a.double() - b).norm() / (b.norm() + 1e-300)).item()
    par = sorted(l2(g[k], ref_g[k]) for k in ref_g if k not in ("bev_query", "feat") and k in g)
    print(f"{label:34s} {l2(out, ref_out):9.2e} {l2(g['bev_query'], ref_g['bev_query']):9.2e} "
          f"{l2(g['feat'], ref_g['feat']):9.2e} {par[-1]:9.2e} {statistics.median(par):9.2e}")


ref_out, ref_g = oracle(torch.float64)
print(f"workload {name}: relative L2 error against the float64 oracle")
print(f"{'':34s} {'output':>9s} {'d query':>9s} {'d feat':>9s} {'d par max':>9s} {'d par med':>9s}")
row("oracle float32 (CPU)", *oracle(torch.float32), ref_out, ref_g)
saved = ops.gemm_mode()
for label, mode, fwd in (("forward GEMMs library fp32", "split", False), ("forward GEMMs split-bf16 MFMA", "split", True),
                         ("forward GEMMs 1-product bf16 MFMA", "bf16", True), ("all GEMMs library fp32 (native)", "native", False)):
    ops.set_gemm_mode(mode)
    with ops.using(train_forward_mfma=fwd):
        row(label, *package(), ref_out, ref_g)
ops.set_gemm_mode(saved)
