"""GPU parity of the hand-written HIP operator (through the C ABI / the
``_ext``-shaped module) against the CPU oracle, forward and backward.

Tolerances (fp32 arithmetic, different summation order than the oracle, and
non-deterministic atomic order in grad_value):
  forward  fp32: rtol 1e-4, atol 1e-5
  backward fp32: rtol 1e-3, atol 1e-4
  bf16 storage : rtol 2e-2, atol 2e-2 against the fp32 oracle fed the same
                 bf16-rounded value (output is rounded to bf16 once more)
"""
import ctypes

import pytest
import torch

from bevformer_amd import _lib
from bevformer_amd import ext
from bevformer_amd import ops
from bevformer_amd.functions import (MultiScaleDeformableAttnFunction_bf16,
                                     MultiScaleDeformableAttnFunction_fp32)
from bevformer_amd.synthetic import make_msda_case, make_sca_msda_case, make_tsa_msda_case
from oracle import bevformer_cpu as O
from oracle import msda_c

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # N, Q, M, D, shapes, P
    (1, 5, 2, 4, [(3, 4)], 2),                              # LPG=1
    (2, 33, 8, 32, [(6, 9), (3, 5)], 4),                    # the encoder's head size
    (6, 70, 8, 32, [(16, 26), (8, 13), (4, 7), (2, 4)], 8),  # SCA-like
    (2, 120, 8, 32, [(12, 10)], 4),                         # TSA-like
    (2, 9, 3, 8, [(5, 7), (3, 4), (2, 2), (1, 1)], 8),      # 1x1 level, M not pow2
    (1, 17, 4, 64, [(7, 5)], 3),                            # generic P, LPG=16
    (1, 6, 1, 256, [(4, 4)], 5),                            # LPG=64
    (1, 4, 2, 5, [(4, 4)], 3),                              # odd D -> scalar fallback
    (2, 11, 2, 16, [(5, 5), (2, 3)], 12),                   # P > LPG
]


def _gpu(*ts):
    return [t.to(DEV) for t in ts]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("coherent", [False, True])
def test_forward_fp32(case, coherent):
    N, Q, M, D, shapes, P = case
    value, sh, start, loc, attn = make_msda_case(N, Q, M, D, shapes, P, seed=1, coherent=coherent)
    want = msda_c.forward(value, sh, start, loc, attn)
    got = ext.ms_deform_attn_forward(*_gpu(value, sh, start, loc, attn), im2col_step=64).cpu()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    # and the grid_sample form (what the reference's CPU fallback computes)
    torch.testing.assert_close(got, O.msda_gridsample(value, sh, loc, attn), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", CASES)
def test_backward_fp32(case):
    N, Q, M, D, shapes, P = case
    value, sh, start, loc, attn = make_msda_case(N, Q, M, D, shapes, P, seed=2)
    g = torch.randn(N, Q, M * D, generator=torch.Generator().manual_seed(9))
    wv, wl, wa = msda_c.backward(value, sh, start, loc, attn, g)
    v, s_, st, l, a, gg = _gpu(value, sh, start, loc, attn, g)
    gv = torch.zeros_like(v)
    gl = torch.full_like(l, 123.0)   # must be overwritten, not accumulated
    ga = torch.full_like(a, 123.0)
    ext.ms_deform_attn_backward(v, s_, st, l, a, gg, gv, gl, ga, im2col_step=64)
    torch.testing.assert_close(gv.cpu(), wv, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(ga.cpu(), wa, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gl.cpu(), wl, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("case", CASES[:7])
def test_forward_backward_bf16(case):
    N, Q, M, D, shapes, P = case
    value, sh, start, loc, attn = make_msda_case(N, Q, M, D, shapes, P, seed=3)
    vb = value.to(torch.bfloat16)
    want = msda_c.forward(vb.float(), sh, start, loc, attn)
    got = ext.ms_deform_attn_forward(*_gpu(vb, sh, start, loc, attn)).float().cpu()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)
    g = torch.randn(N, Q, M * D, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
    wv, wl, wa = msda_c.backward(vb.float(), sh, start, loc, attn, g.float())
    v, s_, st, l, a, gg = _gpu(vb, sh, start, loc, attn, g)
    gv = torch.zeros(v.shape, device=DEV)
    gl = torch.empty_like(l)
    ga = torch.empty_like(a)
    ext.ms_deform_attn_backward(v, s_, st, l, a, gg, gv, gl, ga)
    torch.testing.assert_close(gv.cpu(), wv, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(ga.cpu(), wa, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gl.cpu(), wl, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("rows", [64, 128, 256])
@pytest.mark.parametrize("case", [(3, 300, 8, 32, [(16, 26), (8, 13), (4, 7), (2, 4)], 8),    # SCA-like: 4 levels, 8 points
                                  (2, 1600, 8, 32, [(40, 40)], 4),                            # TSA-like: rows = a 40 x 40 grid
                                  (2, 333, 8, 32, [(9, 11), (4, 6)], 4)])
def test_grad_value_sort_kernel_at_every_workgroup_shape(case, rows):
    """``bevmsda_tuning.reserved[0]`` = 64 / 128 / 256 rows per workgroup of the LDS-sorting grad_value kernel (256 / 512 / 1,024
    threads: 4 / 8 / 16 wavefronts share the stage of finished runs, which overflows the counters' space at 16).  Row counts
    that leave the last workgroup nearly empty: its walk reads the sentinel entries, never what lies behind them (round 6: a
    finished lane group that kept advancing multiplied a zero coefficient with stale LDS words — NaN in one test only)."""
    N, Q, M, D, shapes, P = case
    value, sh, start, loc, attn = make_msda_case(N, Q, M, D, shapes, P, seed=21)
    g = torch.randn(N, Q, M * D, generator=torch.Generator().manual_seed(22))
    wv, wl, wa = msda_c.backward(value, sh, start, loc, attn, g)
    args = _gpu(value, sh, start, loc, attn)
    t = _lib.Tuning()
    t.reserved[0] = rows
    for _ in range(2):
        gv = torch.zeros_like(args[0]); gl = torch.empty_like(args[3]); ga = torch.empty_like(args[4])
        ext.ms_deform_attn_backward(*args, g.to(DEV), gv, gl, ga, tuning=ctypes.byref(t))
        assert torch.isfinite(gv).all()
        torch.testing.assert_close(gv.cpu(), wv, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(ga.cpu(), wa, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gl.cpu(), wl, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("qtile,xcd,variant", [(1, 1, 0), (1, 2, 0), (8, 1, 0), (32, 2, 0),
                                               (128, 2, 0), (8, 2, 2), (8, 2, 1), (1, 2, 1),
                                               (8, 2, 3), (1, 1, 4), (32, 2, 5), (3, 2, 3)])
def test_launch_tunings_agree(qtile, xcd, variant):
    value, sh, start, loc, attn = make_msda_case(3, 77, 8, 32, [(9, 11), (4, 6)], 8, seed=4)
    want = msda_c.forward(value, sh, start, loc, attn)
    t = _lib.Tuning(variant=variant, qtile=qtile, xcd_remap=xcd)
    args = _gpu(value, sh, start, loc, attn)
    got = ext.ms_deform_attn_forward(*args, tuning=ctypes.byref(t)).cpu()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    g = torch.randn(3, 77, 256, generator=torch.Generator().manual_seed(1))
    wv, wl, wa = msda_c.backward(value, sh, start, loc, attn, g)
    gv = torch.zeros_like(args[0]); gl = torch.empty_like(args[3]); ga = torch.empty_like(args[4])
    ext.ms_deform_attn_backward(*args, g.to(DEV), gv, gl, ga, tuning=ctypes.byref(t))
    torch.testing.assert_close(gv.cpu(), wv, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(ga.cpu(), wa, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gl.cpu(), wl, rtol=1e-3, atol=1e-3)


def test_autograd_function_matches_oracle_autograd():
    value, sh, start, loc, attn = make_msda_case(2, 40, 8, 32, [(7, 9), (4, 5)], 4, seed=5)
    g = torch.randn(2, 40, 256, generator=torch.Generator().manual_seed(2))
    wv, wl, wa = O.msda_backward_autograd(value, sh, loc, attn, g)
    v, s_, st, l, a = _gpu(value, sh, start, loc, attn)
    v.requires_grad_(True); l.requires_grad_(True); a.requires_grad_(True)
    out = MultiScaleDeformableAttnFunction_fp32.apply(v, s_, st, l, a, 64)
    out.backward(g.to(DEV))
    torch.testing.assert_close(out.detach().cpu(), O.msda_gridsample(value, sh, loc, attn),
                               rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(v.grad.cpu(), wv, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(a.grad.cpu(), wa, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(l.grad.cpu(), wl, rtol=1e-3, atol=1e-3)
    # fp16 input is up-cast like the reference's custom_fwd(cast_inputs=float32)
    out16 = MultiScaleDeformableAttnFunction_fp32.apply(v.detach().half(), s_, st, l.detach(),
                                                        a.detach(), 64)
    assert out16.dtype == torch.float32
    outb = MultiScaleDeformableAttnFunction_bf16.apply(v.detach(), s_, st, l.detach(), a.detach(), 64)
    assert outb.dtype == torch.bfloat16


def test_empty_and_degenerate_inputs():
    # zero queries
    value, sh, start, loc, attn = make_msda_case(2, 0, 8, 32, [(4, 4)], 4, seed=1)
    out = ext.ms_deform_attn_forward(*_gpu(value, sh, start, loc, attn))
    assert out.shape == (2, 0, 256)
    # every point far outside the map -> exact zeros, gradients zero
    value, sh, start, loc, attn = make_msda_case(1, 9, 8, 32, [(4, 4)], 4, seed=1)
    loc = loc + 5.0
    v, s_, st, l, a = _gpu(value, sh, start, loc, attn)
    out = ext.ms_deform_attn_forward(v, s_, st, l, a)
    assert torch.count_nonzero(out) == 0
    gv = torch.zeros_like(v); gl = torch.full_like(l, 7.0); ga = torch.full_like(a, 7.0)
    ext.ms_deform_attn_backward(v, s_, st, l, a, torch.ones_like(out), gv, gl, ga)
    assert torch.count_nonzero(gv) == 0 and torch.count_nonzero(gl) == 0 and torch.count_nonzero(ga) == 0
    # NaN location contributes nothing (point-level range test fails)
    loc2 = loc - 5.0
    loc2[0, 0, 0, 0, 0, 0] = float("nan")
    out = ext.ms_deform_attn_forward(v, s_, st, loc2.to(DEV), a)
    assert torch.isfinite(out).all()


def test_argument_errors_raise_runtime_error():
    value, sh, start, loc, attn = make_msda_case(1, 4, 2, 8, [(3, 3)], 2, seed=1)
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(value, sh, start, loc, attn)            # CPU tensors
    v, s_, st, l, a = _gpu(value, sh, start, loc, attn)
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(v, s_.int(), st, l, a)                  # wrong index dtype
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(v, s_, st, l.transpose(1, 2), a)        # non-contiguous / shape
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(v.double(), s_, st, l, a)               # unsupported dtype


@pytest.mark.parametrize("which", ["sca", "tsa"])
def test_base_size_properties(which):
    """bevformer_base operator sizes: compared with the (OpenMP) C oracle on a
    slice of queries, plus size-independent properties on the full output:
    linearity in value and the constant-field identity."""
    if which == "sca":
        value, sh, start, loc, attn, _ = make_sca_msda_case("base", seed=0)
    else:
        value, sh, start, loc, attn = make_tsa_msda_case("base", seed=0)
    v, s_, st, l, a = _gpu(value, sh, start, loc, attn)
    out = ext.ms_deform_attn_forward(v, s_, st, l, a)
    # oracle on the first and last 256 queries of every batch entry
    for sl in (slice(0, 256), slice(loc.shape[1] - 256, loc.shape[1])):
        want = msda_c.forward(value, sh, start, loc[:, sl].contiguous(), attn[:, sl].contiguous())
        # fp32 pixel coordinates on a 200-wide map carry ~200 * 2^-24 px of
        # rounding (the oracle forms them in double), hence atol 1e-4 here
        torch.testing.assert_close(out[:, sl].cpu(), want, rtol=1e-4, atol=1e-4)
    # linearity: f(2 v1 - 3 v2) = 2 f(v1) - 3 f(v2)
    v2 = torch.randn(v.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(4))
    out2 = ext.ms_deform_attn_forward(v2, s_, st, l, a)
    mix = ext.ms_deform_attn_forward((2 * v - 3 * v2).contiguous(), s_, st, l, a)
    torch.testing.assert_close(mix, 2 * out - 3 * out2, rtol=1e-4, atol=1e-4)
    # constant field + all taps inside the map: out = const * sum(attn) = const
    lc = l.clamp(0.3, 0.7)
    ones = torch.ones_like(v)
    outc = ext.ms_deform_attn_forward(ones, s_, st, lc, a)
    torch.testing.assert_close(outc, torch.ones_like(outc), rtol=1e-5, atol=1e-5)
    # backward: sum(grad_value) == sum_q g . (sum attn * in-range weights) for constant g
    gv = torch.zeros_like(v); gl = torch.empty_like(l); ga = torch.empty_like(a)
    ext.ms_deform_attn_backward(v, s_, st, lc, a, torch.ones_like(out), gv, gl, ga)
    total = gv.double().sum().item()
    expect = out.shape[0] * out.shape[1] * out.shape[2]       # each output element spreads 1.0
    assert abs(total - expect) / expect < 1e-4


# ----------------------------------------------------------------- fused front end
def _fused_case(kind, seed):
    g = torch.Generator().manual_seed(seed)
    M, D = 8, 32
    if kind == "sca":                      # pillar anchors, ragged rows over 3 cameras
        shapes = [(16, 26), (8, 13), (4, 7), (2, 4)]
        L, P, K, A, N, R, Q = 4, 8, 1, 4, 3, 157, 0
        desc = dict(off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0, vmul=1, vadd=0)
        row_batch = torch.randint(0, N, (R,), generator=g, dtype=torch.int32).sort()[0]
    else:                                  # two BEV-queue entries, averaged
        shapes = [(12, 10)]
        L, P, K, A, R, Q = 1, 4, 2, 1, 2 * 120, 120
        shared = kind == "tsa_shared"
        N = 2 if shared else 4
        desc = dict(off_head=K * L * P * 2, off_k=L * P * 2, lg_head=K * L * P, lg_k=L * P,
                    ref_mode=1, vmul=1 if shared else 2, vadd=0 if shared else 1)
        row_batch = None
    sh = torch.tensor(shapes, dtype=torch.long)
    start = torch.cat([sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]])
    S = int(sh.prod(1).sum())
    value = torch.randn(N, S, M, D, generator=g)
    n_off = M * K * L * P * 2
    proj = torch.randn(R, n_off + M * K * L * P, generator=g)
    proj[:, :n_off] *= 3.0                 # offsets of a few pixels
    ref = torch.rand(R, K, A, 2, generator=g) * 1.2 - 0.1    # some anchors outside the image
    return value, sh, start, proj, n_off, ref, row_batch, dict(M=M, L=L, P=P, K=K, Q=Q, **desc)


@pytest.mark.parametrize("kind", ["sca", "tsa", "tsa_shared"])
def test_fused_front_end_matches_unfused_oracle(kind):
    from bevformer_amd import ops
    from helpers import _oracle_msda_fused
    value, sh, start, proj, n_off, ref, rb, kw = _fused_case(kind, seed=11)
    want = _oracle_msda_fused(value, sh, start, proj, n_off, ref, rb, **kw)
    got = ops.msda_fused(value.to(DEV), sh.to(DEV), start.to(DEV), proj.to(DEV), n_off, ref.to(DEV),
                         rb.to(DEV) if rb is not None else None, **kw)
    assert got is not None
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=2e-5)


def test_fused_front_end_declines_unsupported_shapes():
    from bevformer_amd import ops
    value, sh, start, proj, n_off, ref, rb, kw = _fused_case("tsa", seed=3)
    kw = dict(kw, P=2, off_head=2 * 1 * 2 * 2, off_k=1 * 2 * 2, lg_head=2 * 1 * 2, lg_k=1 * 2)
    n_off2 = 8 * 2 * 1 * 2 * 2
    got = ops.msda_fused(value.to(DEV), sh.to(DEV), start.to(DEV), proj.to(DEV), n_off2, ref.to(DEV),
                         None, **kw)
    assert got is None      # P = 2 is not covered: the caller must use the unfused operator


@pytest.mark.parametrize("kind", ["sca", "tsa"])
def test_fused_front_end_bf16_storage_kernels(kind, monkeypatch):
    """bf16 value storage: the 16-byte-lane kernel (two x-adjacent taps per request, fp32 output
    rows) and the 8-byte-lane kernel against the oracle evaluated on the bf16-rounded value —
    what is left is fp32 accumulation order (and, for the bf16-output kernel, one output
    rounding)."""
    from bevformer_amd import ops
    from helpers import _oracle_msda_fused
    value, sh, start, proj, n_off, ref, rb, kw = _fused_case(kind, seed=12)
    vb = value.to(torch.bfloat16)
    want = _oracle_msda_fused(vb.float(), sh, start, proj, n_off, ref, rb, **kw)
    args = (vb.to(DEV), sh.to(DEV), start.to(DEV), proj.to(DEV), n_off, ref.to(DEV),
            rb.to(DEV) if rb is not None else None)
    ops.set_value_storage(torch.bfloat16)
    try:
        got16 = ops.msda_fused(*args, **kw)
        with ops.using(bf16_lanes8=True):
            got8 = ops.msda_fused(*args, **kw)
    finally:
        ops.set_value_storage(torch.float32)
    assert got16 is not None and got16.dtype == torch.float32
    torch.testing.assert_close(got16.cpu(), want, rtol=1e-4, atol=2e-5)
    assert got8.dtype == torch.bfloat16
    torch.testing.assert_close(got8.float().cpu(), want, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("case", ["sca_shared_rows", "sca_three_levels", "tsa_queue"])
def test_fused_autograd_function_matches_unfused_autograd(case):
    """``ops.msda_fused_autograd`` (fused forward; backward = front-end expand + the operator's backward
    kernels + front-end chain, include/bevmsda.h ``bevmsda_frontend_*``) against autograd through the
    torch statements of the same front end around ``ops.msda_ragged``: output, d/d(value), d/d(projection
    rows) — with several rows sharing a projection row (SCA) and with two queue entries averaged (TSA)."""
    g = torch.Generator().manual_seed(11)
    M, D = 8, 32
    if case.startswith("sca"):
        L, P, K, Dz, Nq, R, N = (4 if case == "sca_shared_rows" else 3), 8, 1, 4, 150, 230, 3
        shapes = torch.tensor([[12, 20], [6, 10], [3, 5], [2, 3]])[:L]
    else:
        L, P, K, Dz, Nq, R, N = 1, 4, 2, 1, 180, 180, 2
        shapes = torch.tensor([[12, 15]])
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    S = int(shapes.prod(1).sum())
    n_off = M * K * L * P * 2
    value = torch.randn(N, S, M, D, generator=g).to(DEV).requires_grad_(True)
    proj = torch.randn(Nq, n_off + M * K * L * P, generator=g)
    proj[:, :n_off] *= 2.0
    proj = proj.to(DEV).requires_grad_(True)
    gout = torch.randn(R, M * D, generator=g).to(DEV)
    sh, st = shapes.to(DEV), start.to(DEV)
    if case.startswith("sca"):
        row_src = torch.randint(0, Nq, (R,), generator=g).to(torch.int32).to(DEV)
        row_batch = torch.randint(0, N, (R,), generator=g).sort().values.to(torch.int32).to(DEV)
        ref = torch.rand(R, 1, Dz, 2, generator=g).to(DEV)
        meta = dict(M=M, L=L, P=P, K=1, off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0, vmul=1, vadd=0)
        out = ops.msda_fused_autograd(value, sh, st, proj, n_off, ref, row_batch, row_src=row_src, **meta)
        out.backward(gout)
        got = (out.detach(), value.grad.clone(), proj.grad.clone())
        value.grad = proj.grad = None
        # the same with the rows of every projection row given: last backward step as a gather (stores)
        from bevformer_amd.modules.geometry import build_q_rows
        out = ops.msda_fused_autograd(value, sh, st, proj, n_off, ref, row_batch, row_src=row_src,
                                      q_rows=build_q_rows(row_src.long(), Nq), **meta)
        out.backward(gout)
        torch.testing.assert_close(proj.grad, got[2], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(value.grad, got[1], rtol=1e-3, atol=1e-4)
        value.grad = proj.grad = None
        # torch statement: point p uses anchor p % Dz (spatial_cross_attention.py:357-372)
        rows = proj[row_src.long()]
        off = rows[:, :n_off].reshape(R, M, L, P, 2)
        att = rows[:, n_off:].reshape(R, M, L * P).softmax(-1).view(R, M, L, P)
        norm = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
        loc = ref.view(R, 1, 1, 1, Dz, 2) + (off / norm[None, None, :, None, :]).view(R, M, L, P // Dz, Dz, 2)
        want = ops.msda_ragged(value, sh, st, loc.reshape(R, M, L, P, 2).contiguous(), att.contiguous(), row_batch)
    else:
        ref = torch.rand(R, K, L, 2, generator=g).to(DEV)
        meta = dict(M=M, L=L, P=P, K=K, off_head=K * L * P * 2, off_k=L * P * 2, lg_head=K * L * P, lg_k=L * P,
                    ref_mode=1, vmul=K, vadd=1, Q=R)
        value = torch.randn(1 * K, S, M, D, generator=g).to(DEV).requires_grad_(True)
        out = ops.msda_fused_autograd(value, sh, st, proj, n_off, ref, None, **meta)
        out.backward(gout)
        got = (out.detach(), value.grad.clone(), proj.grad.clone())
        value.grad = proj.grad = None
        # torch statement: two queue entries with their own value batch entry, averaged
        off = proj[:, :n_off].reshape(R, M, K, L, P, 2).permute(2, 0, 1, 3, 4, 5)          # (K, R, M, L, P, 2)
        att = proj[:, n_off:].reshape(R, M, K, L * P).softmax(-1).view(R, M, K, L, P).permute(2, 0, 1, 3, 4)
        norm = torch.stack([sh[:, 1], sh[:, 0]], -1).float()
        loc = ref.permute(1, 0, 2, 3)[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        rb = torch.arange(K, device=DEV, dtype=torch.int32).repeat_interleave(R)
        o = ops.msda_ragged(value, sh, st, loc.reshape(K * R, M, L, P, 2).contiguous(),
                            att.reshape(K * R, M, L, P).contiguous(), rb)
        want = o.view(K, R, M * D).mean(0)
    want.backward(gout)
    torch.testing.assert_close(got[0], want.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got[1], value.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(got[2], proj.grad, rtol=1e-3, atol=2e-4)
