"""The base SCA backward operator (image-ordered rows, fp32) in a loop, for rocprofv3 --pmc passes over the grad_value
sort kernel: python tools/gv_one.py [iters] [sca_image|sca_raster|tsa]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import ext  # noqa: E402
from bevformer_amd.synthetic import make_sca_msda_case, make_tsa_msda_case  # noqa: E402

DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "sca_image"
if which == "tsa":
    v, sh, st, loc, attn = make_tsa_msda_case("base", seed=0)
else:
    v, sh, st, loc, attn, hits = make_sca_msda_case("base", seed=0)
    if which == "sca_image":
        from bevformer_amd.modules.geometry import _morton_key
        loc_s, attn_s = loc.clone(), attn.clone()
        ctr = loc[:, :, :, 0].mean(dim=(2, 3))
        for i, h in enumerate(hits):
            perm = torch.argsort(_morton_key(ctr[i, :h, 0], ctr[i, :h, 1]), stable=True)
            loc_s[i, :h] = loc[i, :h][perm]
            attn_s[i, :h] = attn[i, :h][perm]
        loc, attn = loc_s, attn_s
vd, shd, std, locd, attnd = v.to(DEV), sh.to(DEV), st.to(DEV), loc.to(DEV), attn.to(DEV)
out = ext.ms_deform_attn_forward(vd, shd, std, locd, attnd)
g = torch.randn_like(out)
gv = torch.zeros(vd.shape, device=DEV)
gl = torch.empty_like(locd)
ga = torch.empty_like(attnd)
for _ in range(iters):
    ext.ms_deform_attn_backward(vd, shd, std, locd, attnd, g, gv, gl, ga)
torch.cuda.synchronize()
print(which, iters, float(gv.abs().mean()))
