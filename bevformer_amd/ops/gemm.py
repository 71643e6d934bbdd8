"""The dense projections: MFMA kernels behind ``linear`` and its fused forms, weight images, ``KERNEL_SELECTION``, the autograd Function."""
import ctypes
import torch

from torch.autograd.function import Function, once_differentiable
from .. import _lib
from ..ext import _ptr
from .. import modes as _modes
from ._base import _NoTimer, _forward_modes, _m
from .sampling import fused_wanted
from .images import _is_transposed_view, packed_weight, panel_weight, transposed_weight


def _pkg():
    """The ``ops`` package: calls between operators go through its namespace, so that a test which substitutes an
    operator there (``tests/helpers.py::oracle_ops``, the ``ops.linear`` spy of the camera-skipping test) sees them too."""
    import sys
    return sys.modules[__package__]


# ---------------------------------------------------------------------------
# Dense projections on the matrix cores (csrc/linear_mfma.h)
# ---------------------------------------------------------------------------
GEMM_MODES = _modes.GEMM_MODES
_GEMM_TIMER = {"cb": None}
# test hook: outputs of launches that may leave row segments unwritten are pre-filled with NaN, so a consumer that
# reads a skipped row cannot go unnoticed (tests/test_frame_plan_gpu.py)
_SEGMENT_POISON = {"on": False, "launches": 0}
_THREAD_STATS = {"inplace": 0, "out_of_place": 0}     # GradThread: how the input gradients of threaded projections were summed


def set_gemm_mode(mode):
    """How the no-grad path runs its nn.Linear layers:
    ``split``  hand-written MFMA kernel, every fp32 operand split into two bf16 terms and
               each product accumulated in fp32 from three bf16 MFMAs (fp32-class result);
    ``bf16``   same kernel, operands rounded to bf16 (one MFMA), fp32 accumulate / output;
    ``native`` torch.nn.functional.linear (hipBLASLt fp32 MFMA at the fp32 vector rate).
    The autograd path always uses ``native``."""
    assert mode in GEMM_MODES
    _modes.process_defaults().gemm = mode


def gemm_mode():
    return _m().gemm


def set_gemm_variant(variant=None, pack=None):
    """Benchmark hook: force a launch variant of the MFMA kernel (None = library default);
    ``pack`` selects the pre-split weight image for the default variant."""
    assert variant in (None, 0, 12), "launch variants: None (default), 0 (fp32 weight matrix), 12 (packed weight image)"
    _modes.process_defaults().gemm_variant = variant
    if pack is not None:
        _modes.process_defaults().gemm_pack = bool(pack)
    elif variant is not None:
        _modes.process_defaults().gemm_pack = variant >= 4


# ---------------------------------------------------------------------------------------------------------------------
# ONE table for every measured threshold that picks a projection kernel.  The Python-side rules read
# their numbers from here; the rules that live in the library (csrc/bevmsda_linear.hip: the library must choose when it
# is called without this package) are listed with the constant that holds them, and tests/test_host_logic_cpu.py checks
# that the two agree and that every profile named exists.
#   rule                       value    what it picks                                          measured                         profile
KERNEL_SELECTION = {
    "panel_min_cols":          (1024,   "row-panel kernel for plain projections with N >= this (the hoisted value projections)",
                                "split mode: camera values 572 (64-row) / 556 (128-row) vs 630 us on the first kernel, BEV values 254 vs 270 us",
                                "profiles/r3/r3a_gemm_ab_first_vs_panel.txt"),
    "panel_two_source_rows":   (16384,  "row-panel kernel (64-row panels) for the K = 512 two-source projection up to this many rows",
                                "a rank's step of an 8-way tiled frame 1.56 -> 1.49 ms (6 launches of 5,000 rows: 19.0 vs 26.1 us isolated, tools/tsa_proj_small_m.py); level from 20,000 rows on",
                                "profiles/r3/r3r_bench_default_sector_tiles_small_m_panel.json"),
    "panel_128_rows":          (1 << 17, "128-row panels (8 wavefronts, half the weight traffic per MFMA) from this many rows on",
                                "camera values (184,950 rows) 556 vs 572 us on 64-row panels; BEV values (80,000 rows) 267 vs 254 us: 64-row panels stay ahead",
                                "profiles/r3/r3a_gemm_ab_first_vs_panel.txt"),
    "layernorm_fused":         (True,   "row-panel kernel whenever the residual + LayerNorm epilogue is wanted (N = 256)",
                                "output_proj + LN 45.4 vs 51.6 us, fc2 + LN 59.3 vs 66.1 us against two launches",
                                "profiles/r3/r3a_gemm_ab_first_vs_panel.txt"),
    "first_64x256_rows":       (32768, "first kernel in 64-row x 256-column tiles (every input row staged once) for 128 < N <= 256 from this many rows on",
                                "TemporalSelfAttention's two-source projection (N = 192, K = 512): 49.1 vs 51.8 us at 40,000 rows, level at 20,000 (32.7 vs 32.8), bit-identical results",
                                "profiles/r6/r6c_first64x256_ab.txt"),
    "tsa_seam":                (True,   "a layer's last chain kernel also makes the NEXT layer's TemporalSelfAttention offset / weight projection (modes.tsa_seam; 32-row workgroups at every row count: the library's rule for bevmsda_proj_ffn_chain_tail_f32)",
                                "chain + tail 127 us (32-row) / 140 us (mixed) vs chain 100 + stand-alone projection 50 us at 40,000 rows; base frame 3.99-4.00 vs 4.05-4.06 ms, bf16 2.86-2.87 vs 2.92 ms",
                                "profiles/r6/r6u_seam_ab2.txt"),
    # ---- in the library (constants of csrc/bevmsda_linear.hip)
    "pipe_max_rows":           (8192,   "kLinearPipeMaxRows: software-pipelined kernel for first-kernel calls of up to this many rows",
                                "13.1-19.9 vs 14.7-23.0 us at 2,500-5,000 rows; level at 10,000, behind from 20,000 rows on",
                                "profiles/r2/r2_gemm_small_m.txt"),
    "chain_small_rows":        (8192,   "kChainSmallRows: 32-row workgroups of the chain kernels up to this many rows",
                                "FFN tail 20.9 vs 28.5 us at 5,000 rows, 33.5 vs 31.1 at 10,000 (the crossover), 56.8 vs 62.0 at 20,000",
                                "profiles/r3/r3k_chain_small_m.txt"),
    "chain_mixed_rows":        (256 * 64, "chain kernels: whole rounds of 64-row workgroups + the tail on 32-row ones from this many rows on",
                                "103.2-103.6 vs 106.6-107.2 us and 100.4-101.5 vs 103.8-105.7 us at 40,000 rows",
                                "profiles/r4/r4f_chain_mixed_shape_ab.txt"),
}


def _sel(name):
    return KERNEL_SELECTION[name][0]


def _panel_covers(N, K0, K1, groups, ln, M=None):
    """Shapes ``bevmsda_linear_panel_f32`` takes (include/bevmsda.h) and, unless a kernel is forced, the ones it is
    faster on (tools/gemm_ab.py, profiles/r3): the hoisted value projections (N >= 1024: 510-570 vs 630 us and 254 vs
    270 us per frame in split mode), the LayerNorm-fused projections (45 vs 52 us, 59 vs 66 us) and the two-source
    projection of TemporalSelfAttention at tile-sized row counts (a rank's share of a BEV-tiled frame: 64-row panels give
    twice the workgroups of 128-row tiles — 19.0 vs 26.1 us at 5,000 rows, level from 20,000 on: tools/tsa_proj_small_m.py);
    the plain per-layer projections (N <= 768, 40 k rows) stay on the first kernel (30-70 us, 5-10 % ahead)."""
    K = K0 + K1
    kern = _m().gemm_kernel
    want = ((ln and _sel("layernorm_fused")) or N >= _sel("panel_min_cols")
            or (K1 > 0 and M is not None and M <= _sel("panel_two_source_rows"))) if kern is None else kern.startswith("panel")
    if not want or _m().gemm_variant is not None or K not in (256, 512) or K0 not in (256, 512) \
            or K1 not in (0, 256) or N % 4:
        return False
    if (K == 512 or ln) and N > 256:
        return False
    if ln and N != 256:
        return False
    return groups == 1 or (N // groups) % 64 == 0


def _panel_call(desc, x0, a0, x1, a1, idx, scale, w, b, ln, y, tag, flops, nbytes, segments=None):
    """One launch of the row-panel kernel; returns False when the library declines the call.  ``segments =
    (seg_start int32 device tensor, seg_len)``: only row segments with entries are computed
    (``bevmsda_linear_panel_segments_f32``); a third entry = the (levels, 2) int64 spatial_shapes of the sampling
    operator that will read the result (its zero-weight taps reach max W + 1 rows into neighbouring segments)."""
    blob = panel_weight(w)
    if blob is None:
        return False
    # panel shape: 128-row panels (half the weight traffic per MFMA, one workgroup per CU) pay from ~128 k rows on
    kern = _m().gemm_kernel or ""
    knob = 0                                                                  # A/B knobs (modes.GEMM_KERNELS)
    for base in ("panel64", "panel128"):
        if kern.startswith(base + "e") and kern[len(base) + 1:].isdigit():
            kern, knob = base, 32 + int(kern[len(base) + 1:])                 # epilogue variant
        elif kern.startswith(base + "s") and kern[len(base) + 1:].isdigit():
            kern, knob = base, 64 + int(kern[len(base) + 1:])                 # phase skew of the column sweep
        elif kern in (base + "d2", base + "d4"):
            kern, knob = base, 97 if kern.endswith("d2") else 98              # one wavefront per SIMD, dripping stores
    if knob and (a0 is not None or x1 is not None or idx is not None or ln is not None):
        knob = 0
    desc.reserved[2] = {"panel64": 1, "panel128": 2, "panel64w2": 1, "panel64w6": 1}.get(kern) \
        or (2 if desc.M >= _sel("panel_128_rows") and ln is None else 1)
    desc.reserved[3] = knob or {"panel64w2": 2, "panel64w6": 6}.get(kern, 0)  # (w2 / w6: weight prefetch depth)
    lib = _lib.load()
    cb = _GEMM_TIMER["cb"]
    ctx = cb(tag, flops, nbytes) if cb is not None else _NoTimer()
    p = lambda t: _ptr(t) if t is not None else None
    with torch.cuda.device(x0.device), ctx:
        if segments is not None and a0 is None and x1 is None and idx is None and ln is None:
            shapes = segments[2] if len(segments) > 2 else None
            if shapes is not None and (shapes.dtype != torch.long or not shapes.is_contiguous() or shapes.dim() != 2):
                raise ValueError("segments[2] must be the contiguous (levels, 2) int64 spatial_shapes tensor")
            rc = lib.bevmsda_linear_panel_segments_f32(p(x0), _ptr(blob), p(b), ctypes.byref(desc), _ptr(segments[0]),
                                                       int(segments[1]), p(shapes), 0 if shapes is None else shapes.shape[0],
                                                       _ptr(y), torch.cuda.current_stream().cuda_stream)
        else:
            rc = lib.bevmsda_linear_panel_f32(p(x0), p(a0), p(x1), p(a1), p(idx), p(scale), _ptr(blob), p(b),
                                              ctypes.byref(desc), ctypes.byref(ln) if ln is not None else None, _ptr(y),
                                              torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED, getattr(_lib, "ERR_TOO_LARGE", -999)):
        return False                    # (TOO_LARGE: an output group beyond the epilogue's 32-bit buffer: the first kernel takes it)
    _lib.check(rc, "linear_panel")
    return True


def set_gemm_kernel(name):
    """Which projection kernel serves the calls several of them cover: ``None`` (by measurement: the row-panel
    kernel for the hoisted N >= 1024 projections and the LayerNorm-fused ones, the software-pipelined kernel for
    M <= 8192, the first kernel otherwise), ``"first"`` (linear_mfma.h), ``"pipe"`` (linear_pipe.h),
    ``"panel"`` / ``"panel64"`` / ``"panel128"`` (linear_panel.h wherever it applies, panel shape by problem
    shape / 64 / 128 rows)."""
    assert name in _modes.GEMM_KERNELS
    _modes.process_defaults().gemm_kernel = name


def gemm_timer_active():
    """Is a GEMM timer registered AND recording?  A registered callback whose owner says ``enabled = False`` (bench.py's
    ``KernelTimer`` between its eager timing passes, e.g. while the step is captured into a HIP graph) brackets nothing — the
    schedule need not keep its launches on one stream for it."""
    cb = _GEMM_TIMER["cb"]
    if cb is None:
        return False
    return bool(getattr(getattr(cb, "__self__", None), "enabled", True))


def set_gemm_timer(cb):
    """``cb(tag, flops, bytes)`` -> context manager around every ``linear`` launch."""
    _GEMM_TIMER["cb"] = cb


def _rows2d(t, K):
    """View ``t`` (..., K) as (rows, K) with unit column stride and one row stride, without
    copying when the leading dims collapse; returns (2-D view, row stride)."""
    if t.dim() != 2:
        t = t.reshape(-1, K)
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) % 4 != 0) or t.data_ptr() % 16 != 0:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else K)


def linear(x, weight, bias=None, *, relu=False, x_add=None, x2=None, x2_add=None, groups=1,
           out_dtype=torch.float32, tag="linear", _inside_autograd=False, segments=None, accumulate_into=None):
    """``act(cat([x (+ x_add), x2 (+ x2_add)], -1) @ weight.T + bias)`` through
    ``bevmsda_linear_f32`` (include/bevmsda.h).  Returns ``None`` when this call is not
    covered (mode ``native``, autograd needed, CPU / non-fp32 tensors, K not a multiple of
    32) and the caller then runs the torch ops.

    ``groups = G > 1``: ``weight`` is the row-wise concatenation of G Linear layers that share
    the input; the result is ``(G, ..., N / G)`` — G contiguous outputs from one pass over x.

    ``segments = (seg_start, seg_len[, spatial_shapes])``: the rows of x are segments of ``seg_len`` rows of which only those with
    ``seg_start[s + 1] > seg_start[s]`` (int32 DEVICE tensor, read by the kernel) will be read by anyone: the others'
    output rows may be left unwritten (row-panel kernel only; other kernels compute everything).

    ``accumulate_into``: an fp32 (rows, N) tensor the result is ADDED to in the kernel's epilogue (and which is
    returned) instead of a fresh output — ``SharedInputGrad``."""
    mode = _m().gemm
    if mode == "native" or not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 \
            or not (_inside_autograd or fused_wanted(x, weight, bias, x_add, x2, x2_add)):
        return None
    K0 = x.shape[-1]
    K1 = x2.shape[-1] if x2 is not None else 0
    N = weight.shape[0]
    if K0 % 32 or K1 % 32 or weight.shape[1] != K0 + K1 or weight.dim() != 2:
        return None
    lead = x.shape[:-1]
    x0, ldx0 = _rows2d(x, K0)
    M = x0.shape[0]
    a0 = a1 = x1 = None
    lda0 = lda1 = ldx1 = 0
    if x_add is not None:
        if x_add.shape != x.shape or x_add.dtype != torch.float32:
            return None
        a0, lda0 = _rows2d(x_add, K0)
    if x2 is not None:
        if x2.shape[:-1] != lead or x2.dtype != torch.float32:
            return None
        x1, ldx1 = _rows2d(x2, K1)
        if x2_add is not None:
            if x2_add.shape != x2.shape or x2_add.dtype != torch.float32:
                return None
            a1, lda1 = _rows2d(x2_add, K1)
    # a transposed VIEW of a row-major matrix (``transposed_weight``: the operand of an input-gradient GEMM) stays a view
    # when its weight image can be packed straight from it (first kernel over the packed image)
    tview = _is_transposed_view(weight) and _m().gemm_pack and _m().gemm_variant is None \
        and not _panel_covers(N, K0, K1, groups, False, M)
    if tview:
        w = weight
    else:
        w = weight if (weight.stride(1) == 1 and weight.stride(0) % 4 == 0
                       and weight.data_ptr() % 16 == 0) else weight.contiguous()
    b = None
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != N:
            return None
        b = bias.contiguous()
    if groups > 1 and (N % groups or (N // groups) % 128):
        return None
    ncol = N // groups
    if out_dtype not in (torch.float32, torch.bfloat16) or (out_dtype == torch.bfloat16 and N % 4):
        return None
    if accumulate_into is not None:
        if groups != 1 or out_dtype != torch.float32 or relu or accumulate_into.dtype != torch.float32 \
                or tuple(accumulate_into.shape) != (M, N) or not accumulate_into.is_contiguous() or N % 4 \
                or accumulate_into.data_ptr() % 16:
            return None
        y = accumulate_into.view(1, M, N)
    else:
        y = torch.empty((groups, M, ncol), dtype=out_dtype, device=x.device)
    if segments is not None and _SEGMENT_POISON["on"]:
        y.fill_(float("nan"))
        _SEGMENT_POISON["launches"] += 1
    if M == 0 or N == 0:
        return y.view(groups, *lead, ncol) if groups > 1 else y.view(*lead, N)
    desc = _lib.LinearDesc(M=M, ldx0=ldx0, lda0=lda0, ldx1=ldx1, lda1=lda1, ldw=w.stride(0),
                           ldy=ncol, N=N, K0=K0, K1=K1, relu=int(bool(relu)),
                           precision=0 if mode == "split" else 1,
                           group_cols=ncol if groups > 1 else 0,
                           out_bf16=int(out_dtype == torch.bfloat16))
    if accumulate_into is not None:
        desc.reserved[0] = 1                # y += result (first kernel)
    if accumulate_into is None and _m().gemm_pack and _panel_covers(N, K0, K1, groups, False, M):
        nbytes = 4 * (M * (K0 + K1) * (1 + (a0 is not None)) + N * (K0 + K1) + M * N)
        if _panel_call(desc, x0, a0, x1, a1, None, None, w, b, None, y, tag, 2.0 * M * N * (K0 + K1), nbytes,
                       segments=segments):
            return y.view(groups, *lead, ncol) if groups > 1 else y.view(*lead, N)
    variant = _m().gemm_variant
    blob = packed_weight(w) if _m().gemm_pack and (variant is None or variant >= 4) else None
    if tview and blob is None:
        w = weight.contiguous()
        desc.ldw = w.stride(0)
    if variant is not None and (variant >= 4) == (blob is not None):
        desc.variant = 1 + variant
    elif blob is not None and _m().gemm_kernel == "pipe" and accumulate_into is None:
        if a0 is None and a1 is None and (K0 + K1) // 32 in (8, 16):
            desc.variant = 131              # force the software-pipelined kernel
    elif _m().gemm_kernel == "first":
        desc.reserved[1] = 1                # keep the first kernel
    elif blob is not None and accumulate_into is None and 128 < N <= 256 and groups == 1 \
            and (_m().gemm_kernel == "first64" or (_m().gemm_kernel is None and _sel("first_64x256_rows") <= M)):
        desc.variant = 17                   # 64 x 256 tiles: every input row staged once (KERNEL_SELECTION)
    lib = _lib.load()
    fn = lib.bevmsda_linear_f32 if blob is None else lib.bevmsda_linear_packed_f32
    cb = _GEMM_TIMER["cb"]
    if cb is not None:
        nbytes = 4 * (M * (K0 + K1) * (1 + (a0 is not None)) + N * (K0 + K1) + M * N)
        ctx = cb(tag, 2.0 * M * N * (K0 + K1), nbytes)
    else:
        ctx = _NoTimer()
    with torch.cuda.device(x.device), ctx:
        rc = fn(_ptr(x0), _ptr(a0) if a0 is not None else None,
                _ptr(x1) if x1 is not None else None, _ptr(a1) if a1 is not None else None,
                _ptr(w) if blob is None else _ptr(blob), _ptr(b) if b is not None else None,
                ctypes.byref(desc), _ptr(y), torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "linear")
    return y.view(groups, *lead, ncol) if groups > 1 else y.view(*lead, N)


def linear_rows2(x_lo, x_hi, weight, bias=None, *, groups=1, out_dtype=torch.float32, tag="linear"):
    """``linear(cat([x_lo, x_hi], 0), ...)`` with the two row blocks read where they lie
    (``bevmsda_linear_panel_rows2_f32``): x_lo (M0, K), x_hi (M1, K) fp32, K = 256 — TSA's value
    ``stack([prev_bev, bev_query])`` projected without forming the stack.  Returns (groups, M0 + M1, N / groups), or
    ``None`` when not covered (the caller stacks and calls ``linear``)."""
    mode = _m().gemm
    if mode == "native" or not x_lo.is_cuda or torch.is_grad_enabled() and (x_lo.requires_grad or x_hi.requires_grad
                                                                            or weight.requires_grad):
        return None
    if x_lo.dtype != torch.float32 or x_hi.dtype != torch.float32 or weight.dtype != torch.float32:
        return None
    N, K = weight.shape
    # (what the row-panel kernel TAKES, not where it is the faster GEMM: the stack it saves outweighs the few per cent the first
    # kernel is ahead at N < 1024)
    if x_lo.shape[-1] != K or x_hi.shape[-1] != K or not _m().gemm_pack or _m().gemm_variant is not None \
            or _m().gemm_kernel in ("first", "pipe") or K != 256 or N % groups or (N // groups) % 128:
        return None
    lo, ld0 = _rows2d(x_lo, K)
    hi, ld1 = _rows2d(x_hi, K)
    if ld0 != ld1:
        return None
    M0, M1 = lo.shape[0], hi.shape[0]
    M, ncol = M0 + M1, N // groups
    if M0 == 0 or M1 == 0 or out_dtype not in (torch.float32, torch.bfloat16):
        return None
    w = weight if (weight.stride(1) == 1 and weight.stride(0) % 4 == 0 and weight.data_ptr() % 16 == 0) else weight.contiguous()
    b = None
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != N:
            return None
        b = bias.contiguous()
    blob = panel_weight(w)
    if blob is None:
        return None
    y = torch.empty((groups, M, ncol), dtype=out_dtype, device=x_lo.device)
    desc = _lib.LinearDesc(M=M, ldx0=ld0, lda0=0, ldx1=0, lda1=0, ldw=w.stride(0), ldy=ncol, N=N, K0=K, K1=0, relu=0,
                           precision=0 if mode == "split" else 1, group_cols=ncol if groups > 1 else 0,
                           out_bf16=int(out_dtype == torch.bfloat16))
    desc.reserved[2] = {"panel64": 1, "panel128": 2, "panel64w2": 1, "panel64w6": 1}.get(_m().gemm_kernel) \
        or (2 if M >= _sel("panel_128_rows") else 1)
    cb = _GEMM_TIMER["cb"]
    ctx = cb(tag, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)) if cb is not None else _NoTimer()
    with torch.cuda.device(x_lo.device), ctx:
        rc = _lib.load().bevmsda_linear_panel_rows2_f32(_ptr(lo), _ptr(hi), M0, _ptr(blob), _ptr(b) if b is not None else None,
                                                        ctypes.byref(desc), _ptr(y), torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "linear_rows2")
    return y


def linear_gather_mean(rows, idx, scale, weight, bias=None, *, tag="linear"):
    """``linear(gather_mean(rows, idx, scale), weight, bias)`` in one kernel
    (``bevmsda_linear_gather_packed_f32``): the camera mean of SpatialCrossAttention folded into
    the A-load of its output projection.  idx (Q, 2) int32.  Returns ``None`` when not covered
    (GEMM mode ``native``, packing off, autograd, more than two cameras per query)."""
    mode = _m().gemm
    if mode == "native" or not _m().gemm_pack or _m().gemm_variant is not None \
            or not rows.is_cuda or rows.dtype != torch.float32 or weight.dtype != torch.float32 \
            or not fused_wanted(rows, weight, bias) or idx.dim() != 2 or idx.shape[1] != 2 \
            or idx.dtype != torch.int32 or rows.dim() != 2 or rows.shape[1] % 32 \
            or weight.shape[1] != rows.shape[1]:
        return None
    rows = rows if rows.stride(1) == 1 and rows.stride(0) % 4 == 0 else rows.contiguous()
    idx = idx.contiguous()
    scale = scale.reshape(-1).float().contiguous()
    Qn, N, K = idx.shape[0], weight.shape[0], rows.shape[1]
    if scale.numel() != Qn:
        return None
    w = weight if (weight.stride(1) == 1 and weight.stride(0) % 4 == 0
                   and weight.data_ptr() % 16 == 0) else weight.contiguous()
    b = bias.contiguous() if bias is not None else None
    y = torch.empty((Qn, N), dtype=torch.float32, device=rows.device)
    if Qn == 0:
        return y
    desc = _lib.LinearDesc(M=Qn, ldx0=rows.stride(0), ldw=K, ldy=N, N=N, K0=K, K1=0, relu=0,
                           precision=0 if mode == "split" else 1)
    if _panel_covers(N, K, 0, 1, False) and rows.data_ptr() % 16 == 0 and _panel_call(
            desc, rows, None, None, None, idx, scale, w, b, None, y, tag, 2.0 * Qn * N * K,
            4.0 * (min(rows.shape[0], 2 * Qn) * K + N * K + Qn * N)):
        return y
    blob = packed_weight(w)
    if blob is None:
        return None
    lib = _lib.load()
    cb = _GEMM_TIMER["cb"]
    # algorithmic bytes: at most two source rows per output row (not the capacity of `rows`)
    ctx = cb(tag, 2.0 * Qn * N * K, 4.0 * (min(rows.shape[0], 2 * Qn) * K + N * K + Qn * N)) if cb is not None else _NoTimer()
    with torch.cuda.device(rows.device), ctx:
        rc = lib.bevmsda_linear_gather_packed_f32(_ptr(rows), rows.stride(0), _ptr(idx), _ptr(scale), _ptr(blob),
                                                  _ptr(b) if b is not None else None, ctypes.byref(desc),
                                                  _ptr(y), torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "linear_gather_mean")
    return y




def set_wgrad_kernel(flag):
    """Weight / bias gradients of the Linear layers on the MFMA kernel (csrc/wgrad_mfma.h; default) or on
    the library's TN GEMM + a column-sum reduction."""
    _modes.process_defaults().wgrad = bool(flag)


def linear_wgrad(g, x, with_bias, *, tag="linear_dw"):
    """(grad_W (N, K), grad_b (N) or None) = (g^T x, g.sum(0)) through ``bevmsda_linear_wgrad_f32``; g (M, N),
    x (M, K) fp32 GPU matrices.  Returns (None, None) when the call is not covered."""
    mode = _m().gemm
    if not _m().wgrad or mode == "native" or not g.is_cuda or g.dtype != torch.float32 \
            or x.dtype != torch.float32 or g.dim() != 2 or x.dim() != 2 or g.shape[0] != x.shape[0]:
        return None, None
    M, N = g.shape
    K = x.shape[1]
    if N % 4 or K % 4 or M == 0:
        return None, None
    g, ldg = _rows2d(g, N)
    x, ldx = _rows2d(x, K)
    # (one zero fill for both accumulators)
    buf = torch.zeros(N * K + (N if with_bias else 0), dtype=torch.float32, device=g.device)
    gw = buf[:N * K].view(N, K)
    gb = buf[N * K:] if with_bias else None
    lib = _lib.load()
    cb = _GEMM_TIMER["cb"]
    ctx = cb(tag, 2.0 * M * N * K, 4.0 * (M * (N + K) + N * K)) if cb is not None else _NoTimer()
    with torch.cuda.device(g.device), ctx:
        rc = lib.bevmsda_linear_wgrad_f32(_ptr(g), ldg, _ptr(x), ldx, M, N, K, _ptr(gw), K,
                                          _ptr(gb) if gb is not None else None, 0 if mode == "split" else 1,
                                          torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None, None
    _lib.check(rc, "linear_wgrad")
    return gw, gb


class GradThread:
    """Several projections of the SAME tensor with their input gradients summed inside the GEMMs.  The camera features go
    through every encoder layer's ``value_proj`` (``[prev_bev, bev_query]`` through every TemporalSelfAttention's):
    autograd would sum the six input gradients with five elementwise adds over a 189 MB (82 MB) tensor each.  Instead the
    tensor is THREADED through the projections: every ``_LinearFunction`` hands out an alias of its input next to its
    result, the next projection consumes the alias, and in backward each function receives the sum so far as the alias'
    gradient, adds its own contribution in its GEMM's epilogue (``linear(accumulate_into=...)``) and passes the same
    buffer on.  Plain autograd data flow: no state outside the graph, a retained graph or a consumer that takes no part
    in the loss need no special case.  ``take(x)`` gives the tensor the next projection must consume, ``put`` stores the
    alias it returned."""
    __slots__ = ("alias",)

    def __init__(self):
        self.alias = None

    def take(self, x):
        a = self.alias                      # (only for the very tensor the thread started from: same memory, same layout)
        if a is None or a.shape != x.shape or a.data_ptr() != x.data_ptr() or a.stride() != x.stride() or a.dtype != x.dtype:
            return x
        return a

    def put(self, alias):
        self.alias = alias


class _LinearFunction(Function):
    """``act(x @ weight.T + bias)`` under autograd, all three GEMMs on this package's MFMA kernels: the
    forward (the projection kernel), the input gradient (``grad_y @ weight``: the projection kernel over the
    transposed weight) and the weight / bias gradient (``grad_y^T x``, a reduction over the rows:
    csrc/wgrad_mfma.h).

    Which GEMM the FORWARD uses matters for the gradients because bilinear sampling is piecewise linear in
    the location: forward round-off moves sampling points across pixel boundaries and single gradient
    entries then take the slope of the other side.  Measured at BASELINE configs[2] against the oracle in
    float64 (tools/train_fwd_table.py, profiles/r2/train_fwd_table.txt), relative L2 error of
    d/d(query), d/d(feat), worst parameter gradient: float32 oracle on the CPU 2.0e-3 / 1.7e-3 / 6.0e-3;
    library fp32 forward 2.0e-3 / 1.7e-3 / 6.0e-3; split-bf16 forward 2.7e-3 / 2.4e-3 / 5.2e-3;
    1-product bf16 forward 4.6e-2 / 3.9e-2 / 7.0e-2.  Any float32 evaluation sits at 2e-3; the split-bf16
    kernel adds a third to that and is the default (``BEVMSDA_TRAIN_FWD_MFMA=0`` restores the library
    forward)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, relu, tag, thread=False):
        ctx.modes = _m().snapshot()
        ctx.thread = bool(thread)
        ctx.set_materialize_grads(False)
        if _m().train_forward_mfma:
            # `weight` itself (not a detached temporary): the packed-weight cache lives on the parameter
            with torch.no_grad():
                y = _pkg().linear(x, weight, bias, relu=relu, tag=tag, _inside_autograd=True)
        else:
            y = torch.nn.functional.linear(x.detach(), weight.detach(), None if bias is None else bias.detach())
            y = torch.relu_(y) if relu else y
        if y is None:                      # shape not covered after all: plain torch, no custom backward
            raise RuntimeError("bevmsda: _LinearFunction called on a shape the MFMA kernel does not cover")
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        ctx.tag = tag
        ctx.save_for_backward(x, weight, y if relu else None)
        if thread:
            return y, x.view_as(x)          # (GradThread: the next projection of x consumes this alias)
        return y

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    @_forward_modes
    def backward(ctx, gy, g_alias=None):
        x, weight, y = ctx.saved_tensors
        K = x.shape[-1]
        if gy is None:                      # my result takes no part in the loss: only the thread passes through
            return (g_alias if ctx.needs_input_grad[0] else None), None, None, None, None, None
        gy = gy.float()
        if ctx.relu:
            gy = torch.ops.aten.threshold_backward(gy.contiguous(), y, 0.0)     # one pass: gy where y > 0
        g2 = gy.reshape(-1, gy.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            acc = None
            # In-place add into the INCOMING gradient of the alias.  Autograd does not promise that an incoming
            # gradient is exclusively ours.  It is when the alias had exactly one consumer — the next projection of the
            # thread, whose backward allocated the buffer, tagged it (`_bevmsda_thread_buffer`) and returned it as its
            # `gx`: the very tensor object arrives here, unsummed and no view of anything.  A second consumer of the
            # alias makes autograd hand over the SUM (a new, untagged tensor), a gradient from anywhere else is
            # untagged too: both take the out-of-place `gx + g_alias` below.  (The alias never leaves GradThread, so
            # no user hook / retain_grad can hold a reference to this buffer.)
            if g_alias is not None and g_alias.dtype == torch.float32 and g_alias.is_contiguous() \
                    and g_alias._base is None and getattr(g_alias, "_bevmsda_thread_buffer", False) \
                    and g_alias.numel() == g2.shape[0] * K:
                acc = g_alias.view(-1, K)   # the sum of the later projections' gradients: add mine in the epilogue
                if _pkg().linear(g2, transposed_weight(weight), None, tag=ctx.tag + "_dx", _inside_autograd=True,
                          accumulate_into=acc) is None:
                    acc = None
            if g_alias is not None:
                _THREAD_STATS["inplace" if acc is not None else "out_of_place"] += 1
            if acc is not None:
                gx = g_alias
            else:
                gx = _pkg().linear(g2, transposed_weight(weight), None, tag=ctx.tag + "_dx", _inside_autograd=True)
                if gx is None:
                    gx = g2 @ weight
                gx = gx.view(x.shape)
                if g_alias is not None:
                    gx = gx + g_alias
            if ctx.thread:
                try:
                    gx._bevmsda_thread_buffer = True      # (a fresh buffer of this thread: the previous projection may add into it)
                except AttributeError:
                    pass
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw, gb_k = linear_wgrad(g2, x.reshape(-1, K), want_b, tag=ctx.tag + "_dw")
            if gw is None:
                gw = g2.t() @ x.reshape(-1, K)
            elif want_b:
                gb = gb_k
        if want_b and gb is None:
            gb = g2.sum(0)
        return gx, gw, gb, None, None, None


def linear_or_torch(x, weight, bias=None, *, relu=False, tag="linear", thread=None):
    """``linear`` with the torch statement as the not-covered path (single-source form).  Under
    autograd (GEMM mode not ``native``) the MFMA kernel runs inside ``_LinearFunction``; ``thread``: the
    ``GradThread`` common to all the projections of this same ``x``."""
    if thread is not None:
        x = thread.take(x)
    y = _pkg().linear(x, weight, bias, relu=relu, tag=tag)
    if y is not None:
        return y
    if _m().gemm != "native" and torch.is_grad_enabled() and not torch.is_autocast_enabled() \
            and x.is_cuda and x.dtype == torch.float32 \
            and weight.dtype == torch.float32 and weight.dim() == 2 and x.shape[-1] % 32 == 0 \
            and weight.shape[0] % 32 == 0 and weight.shape[1] == x.shape[-1] \
            and (x.requires_grad or weight.requires_grad):
        if thread is not None and x.requires_grad:
            y, alias = _LinearFunction.apply(x, weight, bias, relu, tag, True)
            thread.put(alias)
            return y
        return _LinearFunction.apply(x, weight, bias, relu, tag, False)
    y = torch.nn.functional.linear(x, weight, bias)
    if relu:
        y = torch.relu_(y)
    return y
