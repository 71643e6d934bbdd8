"""Rows per workgroup of the grad_value sort kernel (bevmsda_tuning.reserved[0] = 64 / 128 / 256) on the image-ordered base SCA
operator and the TSA operator: whole-backward time by HIP events.  GPU box: python tools/gv_rows_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import _lib, ext  # noqa: E402
from bevformer_amd.synthetic import make_sca_msda_case, make_tsa_msda_case  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = "cuda:0"
v, sh, st, loc, attn, hits = make_sca_msda_case("base", seed=0)
from bevformer_amd.modules.geometry import _morton_key  # noqa: E402
loc_s, attn_s = loc.clone(), attn.clone()
ctr = loc[:, :, :, 0].mean(dim=(2, 3))
for i, h in enumerate(hits):
    perm = torch.argsort(_morton_key(ctr[i, :h, 0], ctr[i, :h, 1]), stable=True)
    loc_s[i, :h] = loc[i, :h][perm]
    attn_s[i, :h] = attn[i, :h][perm]
cases = {"sca_image": (v, sh, st, loc_s, attn_s), "tsa": make_tsa_msda_case("base", seed=0)}
for name, (v, sh, st, loc, attn) in cases.items():
    vd, shd, std, locd, attnd = v.to(DEV), sh.to(DEV), st.to(DEV), loc.to(DEV), attn.to(DEV)
    out = ext.ms_deform_attn_forward(vd, shd, std, locd, attnd)
    g = torch.randn_like(out)
    gv = torch.zeros(vd.shape, device=DEV)
    gl = torch.empty_like(locd)
    ga = torch.empty_like(attnd)
    for rows in (0, 64, 128, 256):
        t = _lib.Tuning()
        t.reserved[0] = rows
        tp = ctypes.byref(t)
        ext.ms_deform_attn_backward(vd, shd, std, locd, attnd, g, gv, gl, ga, tuning=tp)
        med, mn = timeit(lambda: ext.ms_deform_attn_backward(vd, shd, std, locd, attnd, g, gv, gl, ga, tuning=tp), 20)
        print(f"{name} rows_per_workgroup {rows or 'default'}: backward (sort + gather) {med * 1e6:.1f} us (min {mn * 1e6:.1f})", flush=True)
