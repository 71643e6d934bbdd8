import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O
from helpers import build_pair
DEV = torch.device("cuda:0")
for temporal in (False, True):
    enc, sd = build_pair("micro4", device=DEV)
    q, f, kw = S.make_inputs("micro4", seed=2, temporal=temporal)
    kd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    qd, fd = q.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
    out = enc(qd, fd, fd, **kd)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    out.backward(gout.to(DEV))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    qc, fc = q.clone().requires_grad_(True), f.clone().requires_grad_(True)
    want = O.encoder_forward(sdg, qc, fc, pc_range=S.PC_RANGE, **kw)
    want.backward(gout)
    rel = lambda a, b: ((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item()
    print(os.environ.get("DBG_FWD_TORCH"), os.environ.get("DBG_DX_TORCH"), os.environ.get("BEVMSDA_GEMM"), "temporal", temporal,
          "out %.1e q %.1e f %.1e" % (rel(out.detach(), want.detach()), rel(qd.grad, qc.grad), rel(fd.grad, fc.grad)),
          [(n, "%.1e" % rel(p.grad, sdg[n].grad)) for n, p in enc.named_parameters() if rel(p.grad, sdg[n].grad) > 2e-3][:6])
