"""Debug: training forward / backward with and without the saving sampling kernel (micro4)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import build_pair
from bevformer_amd import ops, synthetic as S
DEV = torch.device("cuda:0")
enc, _ = build_pair("micro4", device=DEV)
q, f, kw = S.make_inputs("micro4", seed=6, temporal=True, device=DEV)
gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(3)).to(DEV)
def grads():
    enc.zero_grad(set_to_none=True)
    qd, fd = q.clone().requires_grad_(True), f.clone().requires_grad_(True)
    out = enc(qd, fd, fd, **kw)
    out.backward(gout)
    return out.detach(), {"bev_query": qd.grad, "feats": fd.grad, **{k: p.grad for k, p in enc.named_parameters()}}
oa, ga = grads()
oa2, ga2 = grads()
with ops.using(fused_save=False):
    ob, gb = grads()
print("out a vs a (repeat)", (oa - oa2).abs().max().item(), " a vs b", (oa - ob).abs().max().item(), "scale", oa.abs().max().item())
worst = 0
for k in gb:
    d = (ga[k] - gb[k]).norm().item() / (gb[k].norm().item() + 1e-30)
    d2 = (ga[k] - ga2[k]).norm().item() / (ga[k].norm().item() + 1e-30)
    if d > 1e-6 or d2 > 1e-6:
        print(f"{k:70s} a-b {d:.2e}  a-a {d2:.2e}")
