"""Backward operator bench on the GPU box: first-generation (variant 3: one memory-side atomic per
tap) vs LDS-tiled grad_value (default) at the bevformer_base SCA / TSA operator shapes, raster and
image-ordered rows.  usage: python tools/kbench2.py [bwd|fwd] [--iters N]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import _lib, ext  # noqa: E402
from bevformer_amd.synthetic import make_sca_msda_case, make_tsa_msda_case  # noqa: E402
from kbench import timeit, alg_bytes_bwd  # noqa: E402

DEV = "cuda:0"


def main():
    iters = 10
    cases = {}
    v, sh, st, loc, attn, hits = make_sca_msda_case("base", seed=0)
    cases["sca_raster"] = (v, sh, st, loc, attn)
    from bevformer_amd.modules.geometry import _morton_key
    loc_s, attn_s = loc.clone(), attn.clone()
    ctr = loc[:, :, :, 0].mean(dim=(2, 3))
    for i, h in enumerate(hits):
        key = _morton_key(ctr[i, :h, 0], ctr[i, :h, 1])
        perm = torch.argsort(key, stable=True)
        loc_s[i, :h] = loc[i, :h][perm]
        attn_s[i, :h] = attn[i, :h][perm]
    cases["sca_image"] = (v, sh, st, loc_s, attn_s)
    cases["tsa"] = make_tsa_msda_case("base", seed=0)
    res = []
    for name, (v, sh, st, loc, attn) in cases.items():
        for dtype in (torch.float32, torch.bfloat16):
            vd, shd, std, locd, attnd = v.to(DEV, dtype), sh.to(DEV), st.to(DEV), loc.to(DEV), attn.to(DEV)
            out = ext.ms_deform_attn_forward(vd, shd, std, locd, attnd)
            g = torch.randn_like(out)
            outs = {}
            for variant in (3, 0):
                t = _lib.Tuning(variant=variant, qtile=0, xcd_remap=0)
                tp = ctypes.byref(t)
                gv = torch.zeros(vd.shape, device=DEV)
                gl = torch.empty_like(locd)
                ga = torch.empty_like(attnd)
                ext.ms_deform_attn_backward(vd, shd, std, locd, attnd, g, gv, gl, ga, tuning=tp)
                outs[variant] = (gv.clone(), gl.clone(), ga.clone())
                med, mn = timeit(lambda: ext.ms_deform_attn_backward(vd, shd, std, locd, attnd, g, gv, gl, ga, tuning=tp), iters)
                bb = alg_bytes_bwd(vd, locd, attnd, g)
                r = dict(op=name, dtype=str(dtype).split(".")[-1], variant=variant, bwd_us=med * 1e6, bwd_min_us=mn * 1e6,
                         alg_MB=bb / 1e6, alg_GBs=bb / med / 1e9, frac_hbm=bb / med / 8e12)
                print(json.dumps(r), flush=True)
                res.append(r)
            a, b = outs[3], outs[0]
            sc = a[0].abs().max().item()
            print(json.dumps(dict(op=name, dtype=str(dtype).split(".")[-1], check="new vs first-generation",
                                  grad_value_max_rel=((a[0] - b[0]).abs().max().item() / sc),
                                  grad_loc_equal=bool(torch.equal(a[1], b[1])), grad_attn_equal=bool(torch.equal(a[2], b[2])))),
                  flush=True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kbench2.json")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
