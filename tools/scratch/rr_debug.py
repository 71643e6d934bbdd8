import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
M = 256
rows = torch.randn(M, 256, device=DEV)
res = torch.zeros(M, 256, device=DEV)
n0 = torch.nn.LayerNorm(256).to(DEV)
w0 = torch.eye(256, device=DEV)
b0 = torch.zeros(256, device=DEV)
w1 = torch.eye(256, device=DEV)[:64].contiguous()
b1 = torch.zeros(64, device=DEV)
with torch.no_grad(), ops.using(ln_fuse=True, chain_shape=3):
    gx, gp = ops.proj_ln_proj_chain(rows, w0, b0, res, n0, w1, b1)
want = torch.nn.functional.layer_norm(rows, (256,))
err = (gx - want).abs()
print("x max err", err.max().item(), "rows bad", (err.max(1).values > 1e-3).sum().item(), "cols bad", (err.max(0).values > 1e-3).sum().item())
print("row errs (first 40):", [round(v, 3) for v in err.max(1).values[:40].tolist()])
print("col errs by 32-tile:", [round(err[:, 32 * t:32 * t + 32].max().item(), 3) for t in range(8)])
# is gx a column permutation of want?
r = 5
m = (gx[r][:, None] - want[r][None, :]).abs() < 1e-4
print("row5: matches per got-col", m.sum(1)[:16].tolist(), "perm first 16:", [int(m[i].nonzero()[0]) if m[i].any() else -1 for i in range(16)])
ep = (gp - want[:, :64]).abs()
print("p max err", ep.max().item())
