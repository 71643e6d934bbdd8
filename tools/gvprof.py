"""Phase clocks of the grad_value sort kernel (bevmsda_tuning.reserved[1..2] of the library): where a workgroup's
time goes, summed over workgroups (lane 0 of each).  Runs on the GPU box."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import _lib, ext  # noqa: E402
from bevformer_amd.synthetic import make_sca_msda_case, make_tsa_msda_case  # noqa: E402

DEV = "cuda:0"
names = ["grad_out + record loads", "zero + count (2 barriers)", "scan (2 barriers)", "place (1 barrier)",
         "segmented reduce + flush", "share boundaries (walk 8)", "tail"]
def image_sorted(case):
    from bevformer_amd.modules.geometry import _morton_key
    v, sh, st, loc, attn, hits = case
    loc_s, attn_s = loc.clone(), attn.clone()
    ctr = loc[:, :, :, 0].mean(dim=(2, 3))
    for i, h in enumerate(hits):
        perm = torch.argsort(_morton_key(ctr[i, :h, 0], ctr[i, :h, 1]), stable=True)
        loc_s[i, :h] = loc[i, :h][perm]
        attn_s[i, :h] = attn[i, :h][perm]
    return v, sh, st, loc_s, attn_s


for case, (v, sh, st, loc, attn, *_) in (("sca_raster", make_sca_msda_case("base", seed=0)),
                                         ("sca_image", image_sorted(make_sca_msda_case("base", seed=0))),
                                         ("tsa", make_tsa_msda_case("base", seed=0))):
    vd, shd, std, locd, attnd = v.to(DEV), sh.to(DEV), st.to(DEV), loc.to(DEV), attn.to(DEV)
    out = ext.ms_deform_attn_forward(vd, shd, std, locd, attnd)
    g = torch.randn_like(out)
    gv = torch.zeros(vd.shape, device=DEV)
    gl = torch.empty_like(locd)
    ga = torch.empty_like(attnd)
    prof = torch.zeros(8, dtype=torch.int64, device=DEV)
    tun = _lib.Tuning()
    tun.reserved[1], tun.reserved[2] = ctypes.c_int32(prof.data_ptr() & 0xffffffff).value, ctypes.c_int32(prof.data_ptr() >> 32).value
    ext.ms_deform_attn_backward(vd, shd, std, locd, attnd, g, gv, gl, ga, tuning=ctypes.byref(tun))
    torch.cuda.synchronize()
    p = prof.cpu().tolist()
    tot = sum(p)
    print(case, "total clocks (sum over workgroups)", tot)
    print("   flush atomics (lines):", p[7], " taps:", locd.numel() // 2 * 4)
    for n, c in zip(names, p[:7]):
        print(f"   {n:32s} {c:14d}  {100.0 * c / max(tot, 1):5.1f} %")
