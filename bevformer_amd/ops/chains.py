"""The fused row-local forms of the projections: ``linear_layernorm`` (projection + residual + LayerNorm) and the chain kernels of
an encoder layer's seams (``proj_ffn_chain``, ``proj_ln_proj_chain``; csrc/linear_chain.h) with the markers the layer interpreter
reads (``Normed``, ``Chained``, ``NormedWithProj``)."""
import ctypes
import torch

from .. import _lib
from ..ext import _ptr
from .. import modes as _modes
from ._base import _NoTimer, _m
from .sampling import fused_wanted
from .gemm import _GEMM_TIMER, _panel_call, _panel_covers, _rows2d
from .images import panel_weight


class Normed:
    """Marks a module output to which the following "+ identity" and LayerNorm of the encoder layer
    have already been applied (fused into the projection's epilogue)."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


# Residual add + LayerNorm in the epilogue of the projection that precedes them: the row-panel kernel holds complete
# rows per workgroup, so the norm costs one exchange through LDS instead of a second launch over the grid
# (profiles/r3/gemm_ab: output_proj + LN 45.4 vs 51.6 us, fc2 + LN 59.3 vs 66.1 us, 24.7 vs 33.2 us at 5,000 rows).
# Round 2's form of this on 128 x 256 tiles of the first kernel lost to two launches and is retired.


def set_layernorm_fusion(flag):
    """Residual add + LayerNorm in the epilogue of the projection that precedes them (row-panel kernel)."""
    _modes.process_defaults().ln_fuse = bool(flag)


def linear_layernorm(x, weight, bias, res, norm, *, gather=None, tag="linear"):
    """``LayerNorm(linear(A, weight, bias) + res)`` in one kernel (``bevmsda_linear_panel_f32`` with a LayerNorm descriptor),
    A = ``x`` or, with ``gather = (idx (Q, 2) int32, scale (Q,))``, the camera mean of SpatialCrossAttention
    over the rows of ``x``.  ``norm``: an ``nn.LayerNorm`` over N = 256.  Returns ``None`` when not covered
    (then the caller runs the projection and ``add_layernorm``)."""
    mode = _m().gemm
    if not _m().ln_fuse or _m().gemm_kernel in ("first", "pipe") or mode == "native" or not _m().gemm_pack \
            or _m().gemm_variant is not None or not isinstance(norm, torch.nn.LayerNorm) or norm.weight is None or norm.bias is None \
            or not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 \
            or weight.shape[0] != 256 or tuple(norm.normalized_shape) != (256,) \
            or not fused_wanted(x, weight, bias, res, norm.weight):
        return None
    K = weight.shape[1]
    if K % 32 or x.shape[-1] != K:
        return None
    lead = res.shape[:-1] if res is not None else (x.shape[:-1] if gather is None else (gather[0].shape[0],))
    x0, ldx0 = _rows2d(x, K)
    if gather is not None:
        idx, scale = gather
        if idx.dim() != 2 or idx.shape[1] != 2 or idx.dtype != torch.int32:
            return None
        idx = idx.contiguous()
        scale = scale.reshape(-1).float().contiguous()
        M = idx.shape[0]
        if scale.numel() != M:
            return None
    else:
        M = x0.shape[0]
    r2 = None
    ldres = 0
    if res is not None:
        if res.dtype != torch.float32 or res.shape[-1] != 256 or res.numel() != M * 256:
            return None
        r2, ldres = _rows2d(res, 256)
    w = weight if (weight.stride(1) == 1 and weight.stride(0) % 4 == 0 and weight.data_ptr() % 16 == 0) \
        else weight.contiguous()
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != 256):
        return None
    y = torch.empty((M, 256), dtype=torch.float32, device=x.device)
    if M == 0:
        return y.view(*lead, 256)
    desc = _lib.LinearDesc(M=M, ldx0=ldx0, ldw=K, ldy=256, N=256, K0=K, K1=0, relu=0,
                           precision=0 if mode == "split" else 1)
    ln = _lib.LayerNormDesc(res=_ptr(r2) if r2 is not None else None, ldres=ldres, gamma=_ptr(norm.weight),
                            beta=_ptr(norm.bias), eps=float(norm.eps))
    nbytes = 4.0 * ((min(x0.shape[0], 2 * M) if gather is not None else M) * K + 256 * K + M * 256 * (2 if res is not None else 1))
    if _panel_covers(256, K, 0, 1, True) and _panel_call(
            desc, x0, None, None, None, idx if gather is not None else None, scale if gather is not None else None,
            w, bias.contiguous() if bias is not None else None, ln, y, tag, 2.0 * M * 256 * K, nbytes):
        return y.view(*lead, 256)
    return None


class Chained:
    """Marks a module output to which the REST of the layer's row-local chain — "+ identity", norm, FFN,
    "+ identity", norm — has already been applied (``proj_ffn_chain``): the layer skips those steps."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def proj_ffn_chain(rows, weight, bias, res, norm0, fc1, fc2, norm1, *, gather=None, tag="proj_ffn_chain", tail=None):
    """``norm1(x + fc2(relu(fc1(x))))`` with ``x = norm0(linear(A, weight, bias) + res)`` in ONE kernel
    (``bevmsda_proj_ffn_chain_f32``, csrc/linear_chain.h): the attention's output projection, "+ identity", the
    layer's norm, the FFN, "+ identity" and the next norm — every op local to a BEV row.  A = ``rows`` or, with
    ``gather = (idx (M, 2) int32, scale (M,))``, SpatialCrossAttention's camera mean over the rows (idx (M, J > 2): every
    camera's row of a slot, -1 = absent, present ones first — the rare slots with a third.. row are summed in the kernel in the
    order the stand-alone ``fold_extra_rows`` launch would: bit-equal, one launch less).  ``fc1`` / ``fc2``:
    the FFN's ``nn.Linear`` layers (256 -> 512 -> 256), ``norm0`` / ``norm1``: ``nn.LayerNorm(256)``.  Returns
    ``None`` when not covered (the caller runs the steps one by one).

    ``tail = (first, pos, w3, b3)``: the seam to the NEXT layer in the same launch (``bevmsda_proj_ffn_chain_tail_f32``) —
    ``linear(cat([first, y + pos], -1), w3, b3)``, the next TemporalSelfAttention's merged offset / weight projection of the
    rows ``y`` this launch produces (``first`` (M, 256) rows, ``pos`` (M, 256) rows or None, ``w3`` (N3, 512)).  Returns
    ``(y, proj)`` then; a tail the kernel does not cover is dropped and ``(y, None)`` comes back."""
    if tail is not None:
        out = _proj_ffn_chain(rows, weight, bias, res, norm0, fc1, fc2, norm1, gather, tag, tail)
        if out is not None:
            return out
        y = _proj_ffn_chain(rows, weight, bias, res, norm0, fc1, fc2, norm1, gather, tag, None)
        return None if y is None else (y, None)
    return _proj_ffn_chain(rows, weight, bias, res, norm0, fc1, fc2, norm1, gather, tag, None)


def _proj_ffn_chain(rows, weight, bias, res, norm0, fc1, fc2, norm1, gather, tag, tail):
    m = _m()
    if not m.ln_fuse or m.gemm == "native" or not m.gemm_pack or m.gemm_variant is not None \
            or m.gemm_kernel in ("first", "pipe") or not rows.is_cuda or rows.dtype != torch.float32:
        return None
    for norm in (norm0, norm1):
        if not isinstance(norm, torch.nn.LayerNorm) or tuple(norm.normalized_shape) != (256,) or norm.weight is None \
                or norm.bias is None:
            return None
    if not isinstance(fc1, torch.nn.Linear) or not isinstance(fc2, torch.nn.Linear) \
            or tuple(weight.shape) != (256, 256) or tuple(fc1.weight.shape) != (512, 256) \
            or tuple(fc2.weight.shape) != (256, 512) or fc1.bias is None or fc2.bias is None \
            or not fused_wanted(rows, weight, bias, res, fc1.weight, fc2.weight, norm0.weight, norm1.weight):
        return None
    x0, ldx = _rows2d(rows, 256) if rows.shape[-1] == 256 else (None, 0)
    if x0 is None:
        return None
    idx = scale = None
    if gather is not None:
        idx, scale = gather
        if idx.dim() != 2 or not 2 <= idx.shape[1] <= 64 or idx.dtype != torch.int32:
            return None
        idx = idx.contiguous()
        scale = scale.reshape(-1).float().contiguous()
        M = idx.shape[0]
        if scale.numel() != M:
            return None
    else:
        M = x0.shape[0]
    r2, ldres = None, 0
    if res is not None:
        if res.dtype != torch.float32 or res.shape[-1] != 256 or res.numel() != M * 256:
            return None
        r2, ldres = _rows2d(res, 256)
    lead = res.shape[:-1] if res is not None else (M,)
    y = torch.empty((M, 256), dtype=torch.float32, device=rows.device)
    if M == 0:
        return y.view(*lead, 256)
    ws = []
    for w in (weight, fc1.weight, fc2.weight):
        w = w if (w.stride(1) == 1 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0) else w.contiguous()
        blob = panel_weight(w)
        if blob is None:
            return None
        ws.append(blob)
    desc = _lib.ChainDesc(M=M, ld_rows=ldx, ld_res=ldres, ld_y=256, C=256, F=512, precision=0 if m.gemm == "split" else 1,
                          eps0=float(norm0.eps), eps1=float(norm1.eps))
    desc.reserved[1] = m.chain_shape
    if idx is not None and idx.shape[1] > 2:
        desc.reserved[2] = idx.shape[1]     # every camera's row of a slot: columns 2.. are added to the first row's (no fold launch)
    lib = _lib.load()
    cb = _GEMM_TIMER["cb"]
    flops = 2.0 * M * (256 * 256 + 2 * 256 * 512)
    nbytes = 4.0 * ((min(x0.shape[0], 2 * M) if gather is not None else M) * 256 + M * 256 * (2 if res is not None else 1)
                    + 256 * 256 + 2 * 256 * 512)
    tp = None
    if tail is not None:
        first, pos, w3, b3 = tail
        if w3.dim() != 2 or w3.shape[1] != 512 or w3.shape[0] % 64 or w3.shape[0] > 256 or w3.dtype != torch.float32 \
                or first.dtype != torch.float32 or first.shape[-1] != 256 or first.numel() != M * 256 \
                or (pos is not None and (pos.dtype != torch.float32 or pos.shape[-1] != 256 or pos.numel() != M * 256)) \
                or not fused_wanted(first, w3, b3, pos):
            return None
        f2, ldf = _rows2d(first, 256)
        p2, ldp = _rows2d(pos, 256) if pos is not None else (None, 0)
        w3 = w3 if (w3.stride(1) == 1 and w3.stride(0) % 4 == 0 and w3.data_ptr() % 16 == 0) else w3.contiguous()
        blob3 = panel_weight(w3)
        if f2 is None or (pos is not None and p2 is None) or blob3 is None:
            return None
        N3 = w3.shape[0]
        pr = torch.empty((M, N3), dtype=torch.float32, device=rows.device)
        tp = (f2, ldf, p2, ldp, blob3, b3.contiguous() if b3 is not None else None, N3, pr)
        flops += 2.0 * M * 512 * N3
        nbytes += 4.0 * (M * 256 * (2 if pos is not None else 1) + M * N3 + 512 * N3)
    ctx = cb(tag, flops, nbytes) if cb is not None else _NoTimer()
    p = lambda t: _ptr(t) if t is not None else None
    bc = lambda t: t.contiguous() if t is not None else None
    with torch.cuda.device(rows.device), ctx:
        if tp is None:
            rc = lib.bevmsda_proj_ffn_chain_f32(
                _ptr(x0), p(idx), p(scale), _ptr(ws[0]), p(bc(bias)), p(r2), _ptr(norm0.weight), _ptr(norm0.bias),
                _ptr(ws[1]), p(bc(fc1.bias)), _ptr(ws[2]), p(bc(fc2.bias)), _ptr(norm1.weight), _ptr(norm1.bias),
                ctypes.byref(desc), _ptr(y), torch.cuda.current_stream().cuda_stream)
        else:
            f2, ldf, p2, ldp, blob3, b3c, N3, pr = tp
            rc = lib.bevmsda_proj_ffn_chain_tail_f32(
                _ptr(x0), p(idx), p(scale), _ptr(ws[0]), p(bc(bias)), p(r2), _ptr(norm0.weight), _ptr(norm0.bias),
                _ptr(ws[1]), p(bc(fc1.bias)), _ptr(ws[2]), p(bc(fc2.bias)), _ptr(norm1.weight), _ptr(norm1.bias),
                ctypes.byref(desc), _ptr(y), _ptr(f2), ldf, p(p2), ldp, _ptr(blob3), p(b3c), N3, _ptr(pr), N3,
                torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "proj_ffn_chain")
    if tp is not None:
        return y.view(*lead, 256), tp[-1]
    return y.view(*lead, 256)


class NormedWithProj:
    """A module output to which "+ identity" and the layer's norm have been applied (``t``) together with the NEXT
    attention's projection of those rows (``proj``), both from one kernel (``proj_ln_proj_chain``)."""
    __slots__ = ("t", "proj")

    def __init__(self, t, proj):
        self.t, self.proj = t, proj


def proj_ln_proj_chain(rows, weight, bias, res, norm0, w1, b1, *, tag="proj_ln_proj_chain"):
    """``x = norm0(linear(rows, weight, bias) + res)`` and ``p = linear(x, w1, b1)`` in ONE kernel
    (``bevmsda_proj_ln_proj_chain_f32``, csrc/linear_chain.h MODE 1): TemporalSelfAttention's output projection,
    "+ identity", the layer's norm and SpatialCrossAttention's merged offset / weight projection of the result.
    Returns ``(x, p)`` or ``None`` when not covered."""
    m = _m()
    if not m.ln_fuse or m.gemm == "native" or not m.gemm_pack or m.gemm_variant is not None \
            or m.gemm_kernel in ("first", "pipe") or not rows.is_cuda or rows.dtype != torch.float32 \
            or not isinstance(norm0, torch.nn.LayerNorm) or tuple(norm0.normalized_shape) != (256,) \
            or norm0.weight is None or norm0.bias is None or tuple(weight.shape) != (256, 256) or w1.dim() != 2 \
            or w1.shape[1] != 256 or w1.shape[0] % 32 or w1.shape[0] > 768 or rows.shape[-1] != 256 \
            or res is None or res.dtype != torch.float32 or res.shape[-1] != 256 \
            or not fused_wanted(rows, weight, bias, res, norm0.weight, w1, b1):
        return None
    x0, ldx = _rows2d(rows, 256)
    M = x0.shape[0]
    if res.numel() != M * 256:
        return None
    r2, ldres = _rows2d(res, 256)
    N2 = w1.shape[0]
    x = torch.empty((M, 256), dtype=torch.float32, device=rows.device)
    pr = torch.empty((M, N2), dtype=torch.float32, device=rows.device)
    lead = res.shape[:-1]
    if M == 0:
        return x.view(*lead, 256), pr
    blobs = []
    for w in (weight, w1):
        w = w if (w.stride(1) == 1 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0) else w.contiguous()
        blob = panel_weight(w)
        if blob is None:
            return None
        blobs.append(blob)
    desc = _lib.ChainDesc(M=M, ld_rows=ldx, ld_res=ldres, ld_y=256, C=256, F=N2, precision=0 if m.gemm == "split" else 1,
                          eps0=float(norm0.eps), eps1=0.0)
    desc.reserved[0] = N2
    desc.reserved[1] = m.chain_shape
    lib = _lib.load()
    cb = _GEMM_TIMER["cb"]
    ctx = cb(tag, 2.0 * M * 256 * (256 + N2), 4.0 * (M * 256 * 3 + M * N2 + 256 * 256 + N2 * 256)) if cb is not None else _NoTimer()
    p = lambda t: _ptr(t) if t is not None else None
    bc = lambda t: t.contiguous() if t is not None else None
    with torch.cuda.device(rows.device), ctx:
        rc = lib.bevmsda_proj_ln_proj_chain_f32(_ptr(x0), None, None, _ptr(blobs[0]), p(bc(bias)), _ptr(r2), _ptr(norm0.weight),
                                                _ptr(norm0.bias), _ptr(blobs[1]), p(bc(b1)), ctypes.byref(desc), _ptr(x), _ptr(pr),
                                                torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "proj_ln_proj_chain")
    return x.view(*lead, 256), pr
