import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from helpers import build_pair
from bevformer_amd import ops, synthetic as S
DEV = torch.device("cuda:0")
batch = sys.argv[1] == "1"
ops.set_training_image_batching(batch)
enc, _ = build_pair("micro4", device=DEV)
for p in enc.parameters(): p.requires_grad_(True)
q, f, kw = S.make_inputs("micro4", seed=7, temporal=True, device=DEV)
gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(2)).to(DEV)
holder = {}
def step():
    enc.zero_grad(set_to_none=True)
    out = enc(q, f, f, **kw)
    out.backward(gout)
    holder["out"] = out.detach(); holder["g"] = {k: p.grad for k, p in enc.named_parameters()}
def sgd(scale):
    with torch.no_grad():
        for i, p in enumerate(enc.parameters()): p.add_(torch.full_like(p, scale * (1 + i % 3)))
step(); step()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph): step()
sgd(-2e-3)
graph.replay(); torch.cuda.synchronize()
got_out, got = holder["out"].clone(), {k: v.clone() for k, v in holder["g"].items()}
ops.set_training_image_batching(False)
step(); torch.cuda.synchronize()
want_out, want = holder["out"], holder["g"]
print("batching", batch, "output rel err", ((got_out - want_out).norm() / want_out.norm()).item())
worst = sorted((((got[k] - want[k]).norm() / (want[k].norm() + 1e-30)).item(), k) for k in want)[-4:]
print(worst)
