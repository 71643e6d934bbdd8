"""GPU: the operator's backward (second generation: LDS counting sort + segmented reduction for grad_value,
gather kernel for grad_loc / grad_attn; csrc/msda_bwd_lds.h, msda_bwd_gather.h) against the plain-C double
oracle AT THE SIZES THAT ARE BENCHED — the padded bevformer_base SCA operands with image-ordered rows (128-row
workgroups of one head), the dense 200 x 200 TSA operands (16 x 16 grid tiles) and the ragged 45,960-row call —
element-wise, all three gradients, fp32 and bf16 value storage.  The unit-sized cases of tests/test_msda_gpu.py
fit one workgroup; these do not.  Reference: multi_scale_deformable_attn_function.py:130-163.

Tolerances: rtol 1e-3, atol 1e-4 x the tensor's scale.  grad_loc is discontinuous where a sampling
point sits on a pixel boundary (the oracle evaluates floor() in double, the kernel in float): points closer than
1e-4 px to a boundary are excluded and COUNTED; the count must stay below 1e-3 of the points."""
import functools

import pytest
import torch

from bevformer_amd import ext
from bevformer_amd.modules.geometry import _morton_key
from bevformer_amd.synthetic import make_sca_msda_case, make_tsa_msda_case
from oracle import msda_c

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
RTOL, ATOL = 1e-3, 1e-4


@functools.lru_cache(maxsize=None)
def _sca_image_ordered():
    """Padded base SCA operands (6 x 9,886 rows, 4 levels, 8 points) with every camera's rows in image (Morton) order:
    the order the encoder's frame plan produces and the one the 128-row workgroups of the sort kernel are built for."""
    v, sh, st, loc, attn, hits = make_sca_msda_case("base", seed=0)
    loc_s, attn_s = loc.clone(), attn.clone()
    ctr = loc[:, :, :, 0].mean(dim=(2, 3))
    for i, h in enumerate(hits):
        perm = torch.argsort(_morton_key(ctr[i, :h, 0], ctr[i, :h, 1]), stable=True)
        loc_s[i, :h] = loc[i, :h][perm]
        attn_s[i, :h] = attn[i, :h][perm]
    return v, sh, st, loc_s, attn_s, tuple(hits)


@functools.lru_cache(maxsize=None)
def _tsa_dense():
    return make_tsa_msda_case("base", seed=0)


def _near_boundary(loc, shapes):
    """(..., L, P) bool: the sampling point lies within 1e-4 px of a pixel boundary in x or y."""
    wh = shapes.flip(-1).to(loc.dtype)                              # (L, 2) = (W, H)
    px = loc.double() * wh[:, None, :].double() - 0.5               # (..., L, P, 2)
    return ((px - px.round()).abs() < 1e-4).any(-1)


def _check(name, got, want, mask=None):
    got, want = got.double().cpu(), want.double()
    if mask is not None:
        got, want = got[~mask], want[~mask]
    scale = want.abs().max().item()
    err = (got - want).abs()
    bound = ATOL * scale + RTOL * want.abs()
    worst = (err / bound).max().item()
    l2 = ((got - want).norm() / (want.norm() + 1e-300)).item()
    print(f"{name}: max |err| {err.max().item():.3e} (scale {scale:.3e}), worst err / bound {worst:.3f}, rel L2 {l2:.2e}")
    assert worst <= 1.0, f"{name}: err / (atol * scale + rtol * |want|) = {worst:.2f}"


def _run(v, sh, st, loc, attn, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    N, Q, M = loc.shape[:3]
    D = v.shape[-1]
    gout = torch.randn(N, Q, M * D, generator=g)
    vs, gs = v.to(dtype), gout.to(dtype)                            # what the kernels read (bf16: rounded storage)
    want = msda_c.backward(vs.float(), sh, st, loc, attn, gs.float())
    vd, locd, attnd = vs.to(DEV), loc.to(DEV), attn.to(DEV)
    gv = torch.zeros(v.shape, device=DEV)
    gl = torch.full(loc.shape, float("nan"), device=DEV)            # fully overwritten by contract
    ga = torch.full(attn.shape, float("nan"), device=DEV)
    ext.ms_deform_attn_backward(vd, sh.to(DEV), st.to(DEV), locd, attnd, gs.to(DEV), gv, gl, ga)
    return (gv, gl, ga), want


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_at_padded_base_sca_operands_image_ordered(dtype):
    v, sh, st, loc, attn, _ = _sca_image_ordered()
    (gv, gl, ga), (wv, wl, wa) = _run(v, sh, st, loc, attn, dtype, seed=11)
    near = _near_boundary(loc, sh)
    frac = near.float().mean().item()
    print(f"points within 1e-4 px of a pixel boundary: {int(near.sum())} of {near.numel()} ({frac:.2e})")
    assert frac < 1e-3
    _check("grad_value", gv, wv)
    _check("grad_attn", ga, wa, near)
    _check("grad_loc", gl, wl, near[..., None].expand_as(wl))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_at_dense_base_tsa_operands(dtype):
    v, sh, st, loc, attn = _tsa_dense()
    (gv, gl, ga), (wv, wl, wa) = _run(v, sh, st, loc, attn, dtype, seed=12)
    near = _near_boundary(loc, sh)
    assert near.float().mean().item() < 1e-3
    _check("grad_value", gv, wv)
    _check("grad_attn", ga, wa, near)
    _check("grad_loc", gl, wl, near[..., None].expand_as(wl))


def test_backward_ragged_at_the_base_row_count():
    """The ragged call of the encoder (45,960 rows = the visible (camera, query) pairs, no padding rows) through the
    C ABI's ragged entry point against one oracle call per camera."""
    import ctypes
    from bevformer_amd import _lib
    v, sh, st, loc, attn, hits = _sca_image_ordered()
    N, S, M, D = v.shape
    L, P = loc.shape[3:5]
    R = sum(hits)
    assert R == 45960
    rloc = torch.cat([loc[i, :h] for i, h in enumerate(hits)]).contiguous()
    rattn = torch.cat([attn[i, :h] for i, h in enumerate(hits)]).contiguous()
    rb = torch.cat([torch.full((h,), i, dtype=torch.int32) for i, h in enumerate(hits)])
    gout = torch.randn(R, M * D, generator=torch.Generator().manual_seed(13))
    wv = torch.zeros_like(v)
    wl, wa = torch.zeros_like(rloc), torch.zeros_like(rattn)
    r0 = 0
    for i, h in enumerate(hits):
        a, b, c = msda_c.backward(v[i:i + 1], sh, st, rloc[r0:r0 + h][None], rattn[r0:r0 + h][None], gout[r0:r0 + h][None])
        wv[i], wl[r0:r0 + h], wa[r0:r0 + h] = a[0], b[0], c[0]
        r0 += h
    vd, locd, attnd, gd, rbd = v.to(DEV), rloc.to(DEV), rattn.to(DEV), gout.to(DEV), rb.to(DEV)
    shd, std = sh.to(DEV), st.to(DEV)
    gv = torch.zeros(v.shape, device=DEV)
    gl = torch.full(rloc.shape, float("nan"), device=DEV)
    ga = torch.full(rattn.shape, float("nan"), device=DEV)
    rc = _lib.load().bevmsda_backward_ragged_f32(
        vd.data_ptr(), shd.data_ptr(), std.data_ptr(), locd.data_ptr(), attnd.data_ptr(), rbd.data_ptr(), gd.data_ptr(),
        N, S, M, D, L, R, P, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "backward_ragged")
    near = _near_boundary(rloc, sh)
    _check("grad_value", gv, wv)
    _check("grad_attn", ga, wa, near)
    _check("grad_loc", gl, wl, near[..., None].expand_as(wl))
