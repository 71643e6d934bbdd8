"""The autograd path of an encoder layer on the INFERENCE kernels (round 4).

Up to round 3 everything that fuses the row-local part of ``BEVFormerLayer`` (encoder.py:356-404) — the two row-chain
kernels (csrc/linear_chain.h), the grouped value projections — ran under ``torch.no_grad()`` only; with gradients the
layer fell back to one autograd ``Function`` per ``nn.Linear`` / LayerNorm (96 launches of the first-generation GEMM
kernel per frame, 66 fills, 19 adds).  Here the same kernels are the FORWARD of four autograd Functions, each of which
saves what its backward needs (the chain kernels store the LayerNorm inputs and the FFN's hidden rows, which the
inference launch keeps on chip) and runs a hand-ordered backward out of the library's kernels:

  ``grouped_linear``      the layer-invariant value projections of all layers from one pass over their shared input
                          (``value_proj`` of every MSDeformableAttention3D / TemporalSelfAttention: spatial_cross_
                          attention.py:334, temporal_self_attention.py:198); backward: one weight-gradient launch per
                          layer, the input gradients of all layers summed in the GEMM epilogues;
  ``two_source_linear``   TemporalSelfAttention's ``[sampling_offsets ; attention_weights]`` projection of
                          ``cat([value[:bs], query + query_pos], -1)`` read in place from its two sources
                          (temporal_self_attention.py:186-211);
  ``seam_t``              ``output_proj`` + "+ identity" + ``norms[0]`` + the next attention's projection
                          (temporal_self_attention.py:267-272, encoder.py:376-378, spatial_cross_attention.py:338-348);
  ``seam_s``              camera mean + ``output_proj`` + "+ identity" + ``norms[1]`` + FFN + "+ identity" +
                          ``norms[2]`` (spatial_cross_attention.py:165-175, encoder.py:376-404).

Nothing here reads a device value on the host: with a device-side frame plan the ragged row count stays on the device
through forward AND backward (``nrows``), so a whole training step can be captured in a HIP graph.

No Function here writes into a gradient tensor it was handed: where two gradient addends meet in front of a LayerNorm
(a residual branch's and a projection's input gradient) the LayerNorm-backward kernel forms their sum itself
(``bevmsda_add_layernorm_backward2_f32``), and the weight gradients of a seam go out in one multi-problem launch
(``bevmsda_linear_wgrad_multi_f32``) over buffers the Function zeroed itself.
"""
import ctypes

import torch
from torch.autograd.function import Function, once_differentiable

from . import _lib
from . import ops
from .ext import _ptr
from .ops import _forward_modes, _m

_STATS = {"seam_s": 0, "seam_t": 0, "two_source": 0, "grouped": 0}


def stats():
    """Counters of this module's Functions (tests: which path ran)."""
    return dict(_STATS)


def wanted(*tensors):
    """The fast autograd path applies: gradients are being recorded for CUDA fp32 tensors, the MFMA kernels are in use
    and nothing (autocast, the ``train_chain`` switch) asks for the per-op path."""
    m = _m()
    return m.train_chain and m.fused and m.fused_train and m.ln_fuse and m.gemm != "native" and m.gemm_pack \
        and m.gemm_variant is None and m.gemm_kernel not in ("first", "pipe") and m.train_forward_mfma and m.wgrad \
        and torch.is_grad_enabled() and not torch.is_autocast_enabled() \
        and all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors) \
        and any(t is not None and t.requires_grad for t in tensors)


def _prec():
    return 0 if _m().gemm == "split" else 1


def _aligned(w):
    return w if (w.stride(-1) == 1 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0) else w.contiguous()


def _panel_blob(w):
    blob = ops.panel_weight(_aligned(w))        # (the parameter itself: the image is cached on it per version)
    if blob is None:
        raise RuntimeError("bevmsda: no panel image for a weight of shape %s" % (tuple(w.shape),))
    return blob


def _ln_backward(z, gamma, g, eps, out_shape=None, g2=None):
    """(grad of the LayerNorm input, grad_gamma, grad_beta) for ``y = LayerNorm(z)``: rowops.h's backward kernel with
    its residual operand absent.  ``g2``: a second addend of the incoming gradient (``g + g2`` is formed in the kernel:
    a residual branch's gradient and a projection's input gradient meet here without an add pass).  ``out_shape``:
    shape of the returned input gradient."""
    C = z.shape[-1]
    rows = z.numel() // C
    lib = _lib.load()
    gz = torch.empty(out_shape if out_shape is not None else z.shape, dtype=torch.float32, device=z.device)
    parts = int(lib.bevmsda_add_layernorm_backward_partials(rows))
    scratch = torch.empty(max(parts, 1) * 2 * C, dtype=torch.float32, device=z.device)
    gwb = torch.empty(2, C, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(lib.bevmsda_add_layernorm_backward2_f32(
            _ptr(z), None, _ptr(gamma), _ptr(g), _ptr(g2) if g2 is not None else None, float(eps), rows, C, _ptr(gz),
            _ptr(scratch), _ptr(gwb), torch.cuda.current_stream().cuda_stream), "LayerNorm backward")
    return gz, gwb[0], gwb[1]


def _dgrad(g2, weight, tag, acc=None):
    """``g2 @ weight`` (the input gradient of ``y = x weight^T``), added into ``acc`` in the GEMM's epilogue when
    given (and returned), through the projection kernel over the transposed weight."""
    wt = ops.transposed_weight(weight)
    y = ops.linear(g2, wt, None, tag=tag, _inside_autograd=True, accumulate_into=acc)
    if y is None:
        y = g2 @ weight
        if acc is not None:
            acc.add_(y)
            y = acc
    return y


def _dgrad_relu(g2, weight, act, tag, scale=1.0):
    """``scale * (g2 @ weight) * (act > 0)``: the input gradient of the Linear behind a ReLU (and, with ``scale`` =
    1 / (1 - p), behind the Dropout that followed it: ``act`` is then the activation after that dropout) with the
    backward of both in the GEMM's epilogue (``bevmsda_linear_relu_backward_packed_f32``)."""
    wt = ops.transposed_weight(weight)                      # (in_features, out_features)
    blob = ops.packed_weight(wt) if _m().gemm_pack else None
    M, K = g2.shape
    N = wt.shape[0]
    if blob is not None and act.is_contiguous() and tuple(act.shape) == (M, N) and N % 4 == 0 and K % 32 == 0:
        g2c, ldg = ops._rows2d(g2, K)
        y = torch.empty((M, N), dtype=torch.float32, device=g2.device)
        desc = _lib.LinearDesc(M=M, ldx0=ldg, ldw=K, ldy=N, N=N, K0=K, K1=0, relu=0, precision=_prec())
        desc.reserved[1] = 1                                # (the first kernel: its epilogue carries the mask)
        lib = _lib.load()
        cb = ops._GEMM_TIMER["cb"]
        ctx = cb(tag, 2.0 * M * N * K, 4.0 * (M * K + N * K + 2 * M * N)) if cb is not None else ops._NoTimer()
        with torch.cuda.device(g2.device), ctx:
            rc = lib.bevmsda_linear_relu_backward_packed_f32(_ptr(g2c), _ptr(blob), _ptr(act), N, float(scale),
                                                             ctypes.byref(desc), _ptr(y),
                                                             torch.cuda.current_stream().cuda_stream)
        if rc not in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
            _lib.check(rc, "linear_relu_backward")
            return y
    out = torch.ops.aten.threshold_backward(_dgrad(g2, weight, tag), act, 0.0)
    return out if scale == 1.0 else out * scale


def _wgrad_into(g2, x2, gw, gb, tag):
    """gw (N, K view, row stride = its stride(0)) += g2^T x2, gb (N) += column sums of g2 — ``bevmsda_linear_wgrad_f32``
    (accumulating: the caller zeroed gw / gb)."""
    M, N = g2.shape
    K = x2.shape[1]
    g2, ldg = ops._rows2d(g2, N)
    x2, ldx = ops._rows2d(x2, K)
    lib = _lib.load()
    cb = ops._GEMM_TIMER["cb"]
    ctx = cb(tag, 2.0 * M * N * K, 4.0 * (M * (N + K) + N * K)) if cb is not None else ops._NoTimer()
    with torch.cuda.device(g2.device), ctx:
        rc = lib.bevmsda_linear_wgrad_f32(_ptr(g2), ldg, _ptr(x2), ldx, M, N, K, gw.data_ptr(), gw.stride(0),
                                          gb.data_ptr() if gb is not None else None, _prec(),
                                          torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "linear_wgrad")


def _wgrad_multi(problems, tag):
    """Several weight gradients over the same rows in ONE launch (``bevmsda_linear_wgrad_multi_f32``): ``problems`` =
    [(g2 (M, N), x2 (M, K), gw (N, K view), gb (N) or None), ...], accumulating into zeroed gw / gb."""
    if not problems:
        return
    M = problems[0][0].shape[0]
    arr = (_lib.WgradProblem * len(problems))()
    keep = []
    flops = nbytes = 0.0
    for i, (g2, x2, gw, gb) in enumerate(problems):
        N, K = g2.shape[1], x2.shape[1]
        g2, ldg = ops._rows2d(g2, N)
        x2, ldx = ops._rows2d(x2, K)
        keep += [g2, x2]
        assert g2.shape[0] == M and x2.shape[0] == M
        arr[i].g, arr[i].ldg, arr[i].x, arr[i].ldx = g2.data_ptr(), ldg, x2.data_ptr(), ldx
        arr[i].N, arr[i].K, arr[i].grad_w, arr[i].ldgw = N, K, gw.data_ptr(), gw.stride(0)
        arr[i].grad_b = gb.data_ptr() if gb is not None else None
        flops += 2.0 * M * N * K
        nbytes += 4.0 * (M * (N + K) + N * K)
    lib = _lib.load()
    cb = ops._GEMM_TIMER["cb"]
    ctx = cb(tag, flops, nbytes) if cb is not None else ops._NoTimer()
    dev = problems[0][0].device
    with torch.cuda.device(dev), ctx:
        rc = lib.bevmsda_linear_wgrad_multi_f32(arr, len(problems), M, _prec(), _m().wgrad_workgroups, _m().wgrad_variant,
                                                torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "linear_wgrad_multi")


# Parameter-gradient accumulators of one backward pass: the weight-gradient kernels ADD into zeroed buffers.  The encoder's
# forward (which knows the parameter count) creates a ``GradArena`` for its call and puts it into the modes of the calling
# thread (``ops.using(grad_arena=...)``); every Function of the step snapshots those modes in its forward and re-activates
# them in its backward, so the first ``_zeros`` of the backward pass allocates and clears the arena in ONE fill and every
# later call carves its views out of it (20 launches of a few microseconds each per base frame otherwise).  Whatever does
# not fit — or a backward pass over a graph recorded without an arena — gets its own buffer.  No module-global state.
class GradArena:
    __slots__ = ("floats", "buf", "used", "lock")

    def __init__(self, floats):
        import threading
        self.floats = int(floats) + 64
        self.buf = None
        self.used = 0
        self.lock = threading.Lock()

    def carve(self, total, device):
        """``total`` zeroed floats out of the arena, or None (does not fit / other device)."""
        with self.lock:
            if self.buf is None:
                if self.used:                       # spent by an earlier backward pass over the same graph
                    return None
                self.buf = torch.zeros(self.floats, dtype=torch.float32, device=device)
            a = self.buf
            if a.device != torch.device(device) or self.used + total > a.numel():
                return None
            out = a[self.used:self.used + total]
            self.used += total
            if self.used + 1024 > a.numel():        # spent: drop the reference (the views keep the storage alive)
                self.buf = None
            return out


def begin_step(param_floats):
    """The gradient arena for the backward pass of the forward that is about to be recorded."""
    return GradArena(param_floats)


def _zeros(device, *shapes):
    """One zero fill for several accumulators: views of a single buffer (every size a multiple of 4 floats)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    total = sum(sizes)
    buf = None
    arena = _m().grad_arena if _m().use_grad_arena else None
    if arena is not None and total % 4 == 0:
        buf = arena.carve(total, device)
    if buf is None:
        buf = torch.zeros(total, dtype=torch.float32, device=device)
    out, o = [], 0
    for s, n in zip(shapes, sizes):
        out.append(buf[o:o + n].view(*s))
        o += n
    return out


# ---------------------------------------------------------------------------------------------------------------------
def _is_placeholder(g):
    """The autograd return of a consumer that deposited its gradient in a sink: ``_zero_scalar(...).expand(shape)`` — every
    stride 0 (a sum autograd forms with another consumer's gradient is a dense tensor)."""
    return g is None or (g.numel() > 1 and all(st == 0 for st in g.stride()))


class _GroupedLinearFunction(Function):
    """y_l = x w_l^T + b_l for the L row blocks of ``wcat`` — the hoisted value projections — from ONE pass over x
    (``ops.linear(groups=L)``); x = the row-wise concatenation of ``xs``."""

    @staticmethod
    def forward(ctx, wcat, bcat, L, tag, segments, out_dtype, sink, *xs):
        ctx.modes = _m().snapshot()
        x = xs[0] if len(xs) == 1 else torch.cat([t.reshape(-1, t.shape[-1]) for t in xs], 0)
        y = ops.linear(x.detach(), wcat.detach(), bcat.detach(), groups=L, tag=tag, _inside_autograd=True,
                       segments=segments, out_dtype=out_dtype or torch.float32)
        ctx.sink = sink
        if y is None:
            # a shape / dtype the grouped kernel declines: the same numbers from the library GEMM (the backward below
            # only reads x, wcat and the incoming gradients)
            y = torch.nn.functional.linear(x.detach().to(wcat.dtype), wcat.detach(), bcat.detach())
            y = y.view(-1, L, wcat.shape[0] // L).transpose(0, 1).contiguous()
            if out_dtype is not None:
                y = y.to(out_dtype)
        ctx.L, ctx.tag = L, tag
        ctx.x_shapes = [tuple(t.shape) for t in xs]
        ctx.save_for_backward(x, wcat)
        _STATS["grouped"] += 1
        return tuple(y.view(L, -1, y.shape[-1]).unbind(0))

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, *gys):
        x, wcat = ctx.saved_tensors
        L = ctx.L
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        ncol = wcat.shape[0] // L
        need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_x = [ctx.needs_input_grad[7 + i] for i in range(len(ctx.x_shapes))]
        dW, db = _zeros(x.device, (L * ncol, K), (L * ncol,)) if (need_w or need_b) else (None, None)
        # row ranges of x that want a gradient (e.g. only the current BEV of [history ; current])
        offs, o = [], 0
        for s in ctx.x_shapes:
            n = 1
            for d in s[:-1]:
                n *= d
            offs.append((o, o + n))
            o += n
        lo = min((a for (a, _), nd in zip(offs, need_x) if nd), default=None)
        hi = max((b for (_, b), nd in zip(offs, need_x) if nd), default=None)
        dx = None
        wt = None
        if lo is not None:
            wt = wcat.detach().view(L, ncol, K)
        probs = []
        sink = ctx.sink
        arena = sink.take() if isinstance(sink, ValueGradSink) else None
        if arena is not None and tuple(arena.shape) == (M, L * ncol):
            # the consumers wrote their gradients side by side into one (rows, L * ncol) array: ONE input-gradient GEMM
            # with K = L * ncol (the L partial products are summed in its accumulators instead of L - 1 read-modify-write
            # passes over dx) and one weight-gradient problem with N = L * ncol
            for i, g in enumerate(gys):
                dep, sink[i] = sink[i], None
                col = arena[:, i * ncol:(i + 1) * ncol]
                if dep is not None and dep.data_ptr() == col.data_ptr():
                    if not _is_placeholder(g):
                        # a SECOND consumer of this output (an auxiliary loss on the hoisted values): autograd summed its
                        # gradient with the depositing consumer's zero placeholder — it belongs to the column too
                        col.add_(g.reshape(M, ncol))
                    continue
                src = dep if dep is not None else g         # (a consumer that took another path: its own array)
                if src is not None:
                    col.add_(src.reshape(M, ncol))
            if dW is not None:
                _wgrad_multi([(arena, x2, dW, db)], ctx.tag + "_dw")
            if lo is not None:
                dx = _dgrad(arena[lo:hi], wcat.detach(), ctx.tag + "_dx")
            gys = ()
        for i, g in enumerate(gys):
            if sink is not None and sink[i] is not None:
                # the consumer's fp32 gradient (its autograd return is a placeholder; anything autograd added to the
                # placeholder is a second consumer's gradient)
                extra = None if _is_placeholder(g) else g
                g, sink[i] = sink[i], None
                if extra is not None:
                    g = g.reshape(M, ncol).float() + extra.reshape(M, ncol).float()
            if g is None:
                continue
            g2 = g.reshape(M, ncol).float()
            if dW is not None:
                probs.append((g2, x2, dW[i * ncol:(i + 1) * ncol], db[i * ncol:(i + 1) * ncol]))
            if lo is not None:
                dx = _dgrad(g2[lo:hi], wt[i], ctx.tag + "_dx", acc=dx)
        for j in range(0, len(probs), 8):           # the weight gradients of all layers share the rows: one launch
            _wgrad_multi(probs[j:j + 8], ctx.tag + "_dw")
        grads = []
        for (a, b), s, nd in zip(offs, ctx.x_shapes, need_x):
            grads.append(dx[a - lo:b - lo].view(*s) if (nd and dx is not None) else None)
        return (dW if need_w else None, db if need_b else None, None, None, None, None, None, *grads)


def grouped_linear(xs, wcat, bcat, L, tag, segments=None, out_dtype=None, sink=None):
    """tuple of L tensors (rows, N / L): the projections of the row-wise concatenation of ``xs`` (a tensor or a list of
    tensors with the same last dim) by the L row blocks of ``wcat`` / ``bcat``.  ``out_dtype=torch.bfloat16``: the
    results are rounded in the GEMM's epilogue (bf16 value storage of the sampling kernels); ``sink``: a list of L
    slots in which the consumers deposit their fp32 gradients (``ops.msda_fused_autograd(value_sink=(sink, i))``)."""
    if torch.is_tensor(xs):
        xs = [xs]
    return _GroupedLinearFunction.apply(wcat, bcat, L, tag, segments, out_dtype, sink, *xs)


class ValueGradSink(list):
    """L slots in which the consumers of a grouped projection's outputs deposit their fp32 gradients (the autograd
    return of such a consumer is a zero-stride placeholder).  ``arena=True``: the slots are the column blocks of ONE
    zero-filled (rows, L * width) array allocated on first use in a backward pass — ``buffer(slot, shape, device)`` hands
    out block ``slot`` as a strided (N, S, M, D) view (pixel stride L * M * D) that the grad_value kernels accumulate
    into (``grad_value_stride`` of ``bevmsda_backward_rows_*`` / ``_shared_*``).  A SECOND consumer of an output (an
    auxiliary loss on the hoisted values) is supported: its gradient reaches the projection's backward through autograd, summed
    with the placeholder, and is added to the deposited one there (``_is_placeholder``)."""

    def __init__(self, L, arena=True):
        super().__init__([None] * L)
        self.use_arena = bool(arena)
        self.arena = None

    def buffer(self, slot, shape, device):
        if not self.use_arena:
            return None
        N, S, M, D = shape
        width = M * D
        if self.arena is None:
            self.arena = torch.zeros((N * S, len(self) * width), dtype=torch.float32, device=device)
        if tuple(self.arena.shape) != (N * S, len(self) * width):
            return None
        return self.arena[:, slot * width:(slot + 1) * width].view(N, S, M, D)

    def take(self):
        arena, self.arena = self.arena, None
        return arena


# ---------------------------------------------------------------------------------------------------------------------
class _TwoSourceLinearFunction(Function):
    """y = [first | query + pos] w^T + b with the two K = 256 halves read in place (no ``cat``, no separate add)."""

    @staticmethod
    def forward(ctx, first, query, pos, w, b, tag):
        ctx.modes = _m().snapshot()
        y = ops.linear(first.detach(), w.detach(), None if b is None else b.detach(), x2=query.detach(),
                       x2_add=None if pos is None else pos.detach(), tag=tag, _inside_autograd=True)
        if y is None:
            raise RuntimeError("bevmsda: two-source projection not covered")
        ctx.tag = tag
        ctx.has_pos, ctx.has_b = pos is not None, b is not None
        ctx.save_for_backward(first, query, pos if pos is not None else query.new_empty(0), w)
        _STATS["two_source"] += 1
        return y

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, gy):
        first, query, pos, w = ctx.saved_tensors
        K0, K1 = first.shape[-1], query.shape[-1]
        N = w.shape[0]
        g2 = gy.reshape(-1, N).float()
        ni = ctx.needs_input_grad
        wt = ops.transposed_weight(w)                       # (K0 + K1, N): rows [:K0] = w[:, :K0]^T
        d_first = d_q = None
        if ni[0]:
            d_first = ops.linear(g2, wt[:K0], None, tag=ctx.tag + "_dx", _inside_autograd=True)
            d_first = (g2 @ w[:, :K0] if d_first is None else d_first).view(first.shape)
        if ni[1] or (ctx.has_pos and ni[2]):
            d_q = ops.linear(g2, wt[K0:], None, tag=ctx.tag + "_dx", _inside_autograd=True)
            d_q = (g2 @ w[:, K0:] if d_q is None else d_q).view(query.shape)
        dW = db = None
        if ni[3] or (ctx.has_b and ni[4]):
            dW, db = _zeros(w.device, (N, K0 + K1), (N,))
            # (x2 = query + pos: g^T x2 = g^T query + g^T pos — a third problem ADDING into the same block instead of an add pass)
            probs = [(g2, first.reshape(-1, K0), dW[:, :K0], db), (g2, query.reshape(-1, K1), dW[:, K0:], None)]
            if ctx.has_pos:
                probs.append((g2, pos.reshape(-1, K1), dW[:, K0:], None))
            _wgrad_multi(probs, ctx.tag + "_dw")
        d_pos = None
        if ctx.has_pos and ni[2]:
            d_pos = d_q if not ni[1] else d_q.clone()      # (one buffer per consumer: either may be added into)
        return d_first, d_q if ni[1] else None, d_pos, dW if ni[3] else None, db if (ctx.has_b and ni[4]) else None, None


def two_source_linear(first, query, pos, w, b, tag="tsa_offs_attn"):
    return _TwoSourceLinearFunction.apply(first, query, pos, w, b, tag)


# ---------------------------------------------------------------------------------------------------------------------
class _SeamTFunction(Function):
    """x = LayerNorm0(rows w0^T + b0 + res), p = x w1^T + b1 — ``bevmsda_proj_ln_proj_chain_train_f32``."""

    @staticmethod
    def forward(ctx, rows, w0, b0, res, gamma0, beta0, w1, b1, eps0, tag, drop0=None):
        ctx.modes = _m().snapshot()
        m = _m()
        rows2, ldx = ops._rows2d(rows.detach(), 256)
        M = rows2.shape[0]
        if drop0 is not None:
            drop0 = drop0.detach().reshape(M, 256).float().contiguous()
        res2, ldres = ops._rows2d(res.detach(), 256)
        N2 = w1.shape[0]
        dev = rows.device
        x = torch.empty((M, 256), dtype=torch.float32, device=dev)
        pr = torch.empty((M, N2), dtype=torch.float32, device=dev)
        z0 = torch.empty((M, 256), dtype=torch.float32, device=dev)
        desc = _lib.ChainDesc(M=M, ld_rows=ldx, ld_res=ldres, ld_y=256, C=256, F=N2, precision=_prec(), eps0=float(eps0),
                              eps1=0.0)
        desc.reserved[0] = N2
        desc.reserved[1] = m.chain_shape
        lib = _lib.load()
        cb = ops._GEMM_TIMER["cb"]
        tctx = cb(tag, 2.0 * M * 256 * (256 + N2), 4.0 * (M * 256 * 4 + M * N2 + 256 * 256 + N2 * 256)) if cb is not None \
            else ops._NoTimer()
        keep = []

        def p(t):
            if t is None:
                return None
            keep.append(t.detach().contiguous())
            return _ptr(keep[-1])
        with torch.cuda.device(dev), tctx:
            rc = lib.bevmsda_proj_ln_proj_chain_train_f32(
                _ptr(rows2), _ptr(_panel_blob(w0)), p(b0), _ptr(res2), p(gamma0), p(beta0), _ptr(_panel_blob(w1)), p(b1),
                ctypes.byref(desc), _ptr(x), _ptr(pr), _ptr(z0), _ptr(drop0) if drop0 is not None else None,
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "proj_ln_proj_chain (train)")
        ctx.drop0 = drop0
        ctx.eps0, ctx.tag = float(eps0), tag
        ctx.has_b0, ctx.has_b1 = b0 is not None, b1 is not None
        ctx.res_shape = tuple(res.shape)
        ctx.rows_shape = tuple(rows.shape)
        ctx.save_for_backward(rows2, w0, gamma0, w1, z0, x)
        _STATS["seam_t"] += 1
        return x.view(*res.shape), pr

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, gx, gp):
        rows2, w0, gamma0, w1, z0, x = ctx.saved_tensors
        M, N2 = x.shape[0], w1.shape[0]
        dev = x.device
        ni = ctx.needs_input_grad
        dW0, db0, dW1, db1 = _zeros(dev, (256, 256), (256,), (N2, 256), (N2,))
        if gx is not None:
            gx = gx.reshape(M, 256)
            if gx.dtype != torch.float32 or not gx.is_contiguous():
                gx = gx.float().contiguous()
        probs = []
        gp2 = dxp = None
        if gp is not None and N2 % 256 == 0 and N2 <= 768 and _m().chain_backward and ni[0]:
            # ONE kernel for the row-local part (linear_chain.h MODE 3): gp w1 (+ gx) -> LayerNorm backward -> . w0
            gp2, ldp = ops._rows2d(gp.reshape(M, N2).float(), N2)
            if ni[6] or ni[7]:
                probs.append((gp2, x, dW1, db1))
            dzr = torch.empty(ctx.res_shape, dtype=torch.float32, device=dev)
            d_in = torch.empty((M, 256), dtype=torch.float32, device=dev)
            dzp = torch.empty((M, 256), dtype=torch.float32, device=dev) if ctx.drop0 is not None else None
            gb0, = _zeros(dev, (2, 256))
            desc = _lib.ChainDesc(M=M, ld_rows=256, ld_res=256, ld_y=256, C=256, F=N2, precision=_prec(), eps0=ctx.eps0, eps1=0.0)
            desc.reserved[0] = N2
            blobs = [ops.panel_weight(_aligned(w.detach()).t()) for w in (w0, w1)]
            if any(b is None for b in blobs):
                raise RuntimeError("bevmsda: no transposed panel image for the chain backward")
            cb = ops._GEMM_TIMER["cb"]
            tctx = cb(ctx.tag + "_bwd", 2.0 * M * 256 * (256 + N2), 4.0 * M * (N2 + 256 * 4)) if cb is not None else ops._NoTimer()
            with torch.cuda.device(dev), tctx:
                _lib.check(_lib.load().bevmsda_proj_ln_proj_chain_backward_f32(
                    _ptr(gp2), ldp, _ptr(gx) if gx is not None else None, _ptr(z0), _ptr(gamma0.detach().contiguous()),
                    _ptr(blobs[0]), _ptr(blobs[1]), ctypes.byref(desc), _ptr(dzr), _ptr(d_in), _ptr(gb0),
                    _ptr(ctx.drop0) if ctx.drop0 is not None else None, _ptr(dzp) if dzp is not None else None,
                    torch.cuda.current_stream().cuda_stream), "proj_ln_proj_chain backward")
            if ni[1] or ni[2]:
                probs.append((dzp if dzp is not None else dzr.view(M, 256), rows2, dW0, db0))
            _wgrad_multi(probs, ctx.tag + "_dw")
            return (d_in.view(ctx.rows_shape), dW0 if ni[1] else None, db0 if (ctx.has_b0 and ni[2]) else None,
                    dzr if ni[3] else None, gb0[0] if ni[4] else None, gb0[1] if ni[5] else None, dW1 if ni[6] else None,
                    db1 if (ctx.has_b1 and ni[7]) else None, None, None, None)
        if gp is not None:
            gp2 = gp.reshape(M, N2).float()
            if ni[6] or ni[7]:
                probs.append((gp2, x, dW1, db1))
            dxp = _dgrad(gp2, w1, ctx.tag + "_dx1")          # the projection's share of the gradient of x
        if gx is None and dxp is None:
            return (None,) * 11
        # the two addends of d/dx (the residual branch's gradient, the projection's) meet inside the LayerNorm backward
        ga, gb_ = (gx, dxp) if gx is not None else (dxp, None)
        dzr, dg0, dbe0 = _ln_backward(z0, gamma0, ga, ctx.eps0, out_shape=ctx.res_shape, g2=gb_)
        dz0 = dzr.view(M, 256)
        dzp = dz0 if ctx.drop0 is None else dz0 * ctx.drop0      # through the dropout on the projection's output
        if ni[1] or ni[2]:
            probs.append((dzp, rows2, dW0, db0))
        _wgrad_multi(probs, ctx.tag + "_dw")
        d_rows = _dgrad(dzp, w0, ctx.tag + "_dx0").view(ctx.rows_shape) if ni[0] else None
        d_res = dzr if ni[3] else None
        return (d_rows, dW0 if ni[1] else None, db0 if (ctx.has_b0 and ni[2]) else None, d_res, dg0 if ni[4] else None,
                dbe0 if ni[5] else None, dW1 if ni[6] else None, db1 if (ctx.has_b1 and ni[7]) else None, None, None, None)


def dropout_scale(shape, p, device):
    """The scale tensor (0 or 1 / (1 - p)) of one ``nn.Dropout`` call in train() mode, drawn by the framework's own
    ``torch.nn.functional.dropout`` on ones (same generator stream, same call count as the module it stands for)."""
    return torch.nn.functional.dropout(torch.ones(shape, dtype=torch.float32, device=device), p, True)


def seam_t(rows, w0, b0, res, norm0, w1, b1, tag="tsa_out_sca_proj_chain", drop_p=0.0):
    """(x, p) = (norm0(drop(rows w0^T + b0) + res), x w1^T + b1) with gradients, or None when the call is not covered.
    ``drop_p`` > 0: the attention's ``nn.Dropout`` is active (train() mode)."""
    if not (isinstance(norm0, torch.nn.LayerNorm) and tuple(norm0.normalized_shape) == (256,) and norm0.weight is not None
            and norm0.bias is not None and tuple(w0.shape) == (256, 256) and w1.dim() == 2 and w1.shape[1] == 256
            and w1.shape[0] % 64 == 0 and w1.shape[0] <= 768 and rows.shape[-1] == 256 and res is not None
            and res.shape[-1] == 256 and res.numel() == rows.numel()
            and wanted(rows, w0, b0, res, norm0.weight, norm0.bias, w1, b1)):
        return None
    drop0 = dropout_scale(res.shape, drop_p, rows.device) if drop_p > 0 else None
    return _SeamTFunction.apply(rows, w0, b0, res, norm0.weight, norm0.bias, w1, b1, norm0.eps, tag, drop0)


# ---------------------------------------------------------------------------------------------------------------------
class _SeamSFunction(Function):
    """y = LN1(x + relu(x w1^T + b1) w2^T + b2), x = LN0(mean_cameras(rows) w0^T + b0 + res) —
    ``bevmsda_proj_ffn_chain_train_f32``; ``idx`` (M, 2) / ``scale`` (M): the two-row camera gather, ``row_slot`` (R)
    its inverse, ``nrows``: device-side row count or None."""

    @staticmethod
    def forward(ctx, rows, w0, b0, res, gamma0, beta0, w1, b1, w2, b2, gamma1, beta1, idx, scale, row_slot, nrows, fold,
                eps0, eps1, tag, drops=None):
        ctx.modes = _m().snapshot()
        m = _m()
        rows2, ldx = ops._rows2d(rows.detach(), 256)
        M0 = idx.shape[0]
        # drops = (scale tensor of the attention's dropout, of the FFN's hidden dropout, of its output dropout, p_hidden)
        dk = [None, None, None]
        if drops is not None:
            for i, (t, C_) in enumerate(zip(drops[:3], (256, 512, 256))):
                dk[i] = None if t is None else t.detach().reshape(M0, C_).float().contiguous()
        ctx.drop0, ctx.drop1 = dk[0], dk[2]
        ctx.hidden_scale = 1.0 / (1.0 - drops[3]) if (drops is not None and dk[1] is not None) else 1.0
        if fold is not None:
            # rows of a third.. camera folded into the first (in place, on the sampling Function's fresh output: nobody
            # else reads it, and the version saved below is the folded one — what the backward's gather recomputes from)
            ops.fold_extra_rows(rows2, fold[0], fold[1])
        M = idx.shape[0]
        res2, ldres = ops._rows2d(res.detach(), 256)
        dev = rows.device
        scale = scale.reshape(-1).float().contiguous()
        idx = idx.contiguous()
        y = torch.empty((M, 256), dtype=torch.float32, device=dev)
        z0 = torch.empty((M, 256), dtype=torch.float32, device=dev)
        x = torch.empty((M, 256), dtype=torch.float32, device=dev)
        h = torch.empty((M, 512), dtype=torch.float32, device=dev)
        z1 = torch.empty((M, 256), dtype=torch.float32, device=dev)
        desc = _lib.ChainDesc(M=M, ld_rows=ldx, ld_res=ldres, ld_y=256, C=256, F=512, precision=_prec(), eps0=float(eps0),
                              eps1=float(eps1))
        desc.reserved[1] = m.chain_shape
        lib = _lib.load()
        cb = ops._GEMM_TIMER["cb"]
        flops = 2.0 * M * (256 * 256 + 2 * 256 * 512)
        nbytes = 4.0 * (min(rows2.shape[0], 2 * M) * 256 + M * 256 * 6 + M * 512 + 256 * 256 + 2 * 256 * 512)
        tctx = cb(tag, flops, nbytes) if cb is not None else ops._NoTimer()
        keep = []

        def p(t):
            if t is None:
                return None
            keep.append(t.detach().contiguous())
            return _ptr(keep[-1])
        with torch.cuda.device(dev), tctx:
            rc = lib.bevmsda_proj_ffn_chain_train_f32(
                _ptr(rows2), _ptr(idx), _ptr(scale), _ptr(_panel_blob(w0)), p(b0), _ptr(res2), p(gamma0), p(beta0),
                _ptr(_panel_blob(w1)), p(b1), _ptr(_panel_blob(w2)), p(b2), p(gamma1), p(beta1), ctypes.byref(desc), _ptr(y),
                _ptr(z0), _ptr(x), _ptr(h), _ptr(z1), *[_ptr(t) if t is not None else None for t in dk],
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "proj_ffn_chain (train)")
        ctx.eps0, ctx.eps1, ctx.tag = float(eps0), float(eps1), tag
        ctx.has_b = (b0 is not None, b1 is not None, b2 is not None)
        ctx.res_shape = tuple(res.shape)
        ctx.rows_shape = tuple(rows.shape)
        ctx.has_nrows = nrows is not None
        ctx.save_for_backward(rows2, w0, gamma0, w1, w2, gamma1, idx, scale, row_slot,
                              nrows if nrows is not None else row_slot.new_empty(0), z0, x, h, z1)
        _STATS["seam_s"] += 1
        return y.view(*res.shape)

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, gy):
        rows2, w0, gamma0, w1, w2, gamma1, idx, scale, row_slot, nrows, z0, x, h, z1 = ctx.saved_tensors
        M = x.shape[0]
        dev = x.device
        ni = ctx.needs_input_grad
        tag = ctx.tag
        gy = gy.reshape(M, 256)
        if gy.dtype != torch.float32 or not gy.is_contiguous():
            gy = gy.float().contiguous()
        dW0, db0, dW1, db1, dW2, db2 = _zeros(dev, (256, 256), (256,), (512, 256), (512,), (256, 512), (256,))
        dg = None
        if _m().chain_backward:
            # ONE kernel for the row-local part of the backward (linear_chain.h MODE 2): both LayerNorm backwards, the
            # ReLU-masked FFN input gradients and the output projection's, with the rows resident as in the forward
            dz1 = torch.empty((M, 256), dtype=torch.float32, device=dev)
            dh = torch.empty((M, 512), dtype=torch.float32, device=dev)
            dzr = torch.empty(ctx.res_shape, dtype=torch.float32, device=dev)
            dg = torch.empty((M, 256), dtype=torch.float32, device=dev)
            dzp = torch.empty((M, 256), dtype=torch.float32, device=dev) if ctx.drop0 is not None else None
            gb1, gb0 = _zeros(dev, (2, 256), (2, 256))
            desc = _lib.ChainDesc(M=M, ld_rows=256, ld_res=256, ld_y=256, C=256, F=512, precision=_prec(), eps0=ctx.eps0,
                                  eps1=ctx.eps1)
            blobs = [ops.panel_weight(_aligned(w.detach()).t()) for w in (w0, w1, w2)]      # images of w^T, from w
            if any(b is None for b in blobs):
                raise RuntimeError("bevmsda: no transposed panel image for the chain backward")
            cb = ops._GEMM_TIMER["cb"]
            tctx = cb(tag + "_bwd", 2.0 * M * (256 * 256 + 2 * 256 * 512), 4.0 * M * 256 * 10) if cb is not None else ops._NoTimer()
            with torch.cuda.device(dev), tctx:
                _lib.check(_lib.load().bevmsda_proj_ffn_chain_backward_f32(
                    _ptr(gy), 256, _ptr(z0), _ptr(h), _ptr(z1), _ptr(gamma0.detach().contiguous()),
                    _ptr(gamma1.detach().contiguous()), _ptr(blobs[0]), _ptr(blobs[1]), _ptr(blobs[2]), ctypes.byref(desc),
                    _ptr(dz1), _ptr(dh), _ptr(dzr), _ptr(dg), _ptr(gb1), _ptr(gb0),
                    _ptr(ctx.drop0) if ctx.drop0 is not None else None, _ptr(ctx.drop1) if ctx.drop1 is not None else None,
                    float(ctx.hidden_scale), _ptr(dzp) if dzp is not None else None,
                    torch.cuda.current_stream().cuda_stream), "proj_ffn_chain backward")
            dg1, dbe1, dg0, dbe0 = gb1[0], gb1[1], gb0[0], gb0[1]
            df, dz0 = dz1, dzr.view(M, 256)            # (train() mode: dz1 already carries the FFN-output dropout's scale)
            if dzp is None:
                dzp = dz0
        else:
            # LayerNorm1, FFN
            dz1, dg1, dbe1 = _ln_backward(z1, gamma1, gy, ctx.eps1)
            df = dz1 if ctx.drop1 is None else dz1 * ctx.drop1                     # through the FFN's output dropout
            # (dz1 w2) where h > 0 (h: after its dropout — zero where dropped), times 1 / (1 - p) of that dropout
            dh = _dgrad_relu(df, w2, h, tag + "_dx2", ctx.hidden_scale)            # (M, 512)
            dxf = _dgrad(dh, w1, tag + "_dx1")                                    # the FFN's share of the gradient of x
            # LayerNorm0 (d/dx = the residual branch's dz1 + the FFN's share: added inside the kernel), output projection
            dzr, dg0, dbe0 = _ln_backward(z0, gamma0, dz1, ctx.eps0, out_shape=ctx.res_shape, g2=dxf)
            dz0 = dzr.view(M, 256)
            dzp = dz0 if ctx.drop0 is None else dz0 * ctx.drop0                    # through the attention's dropout
        probs = [(df, h, dW2, db2), (dh, x, dW1, db1)]
        d_rows = None
        if ni[0] or ni[1] or ni[2]:
            g = ops.gather_mean(rows2 if rows2.is_contiguous() else rows2.contiguous(), idx, scale)   # recomputed: (M, 256)
            probs.append((dzp, g, dW0, db0))
        _wgrad_multi(probs, tag + "_dw")            # the three weight gradients of the seam: one launch
        if ni[0]:
            if dg is None:
                dg = _dgrad(dzp, w0, tag + "_dx0")
            R = rows2.shape[0]
            d_rows = torch.empty((R, 256), dtype=torch.float32, device=dev)
            lib = _lib.load()
            with torch.cuda.device(dev):
                _lib.check(lib.bevmsda_rows_from_slots_f32(
                    _ptr(dg), 256, _ptr(scale), _ptr(row_slot), nrows.data_ptr() if ctx.has_nrows else None, R, 256,
                    _ptr(d_rows), torch.cuda.current_stream().cuda_stream), "rows_from_slots")
        d_res = dzr if ni[3] else None
        hb = ctx.has_b
        return (d_rows, dW0 if ni[1] else None, db0 if (hb[0] and ni[2]) else None, d_res, dg0 if ni[4] else None,
                dbe0 if ni[5] else None, dW1 if ni[6] else None, db1 if (hb[1] and ni[7]) else None,
                dW2 if ni[8] else None, db2 if (hb[2] and ni[9]) else None, dg1 if ni[10] else None,
                dbe1 if ni[11] else None, None, None, None, None, None, None, None, None, None)


def seam_s(rows, w0, b0, res, norm0, fc1, fc2, norm1, *, gather, row_slot, nrows=None, fold=None,
           tag="sca_out_ffn_chain", drop_p=(0.0, 0.0, 0.0)):
    """The SCA seam with gradients, or None when not covered.  ``gather = (idx (M, 2) int32, scale (M))``; ``row_slot``
    (R,) int32: the BEV query of every ragged row; ``nrows``: device-side row count of a dynamic frame plan;
    ``fold = (q_rows_all, n_extra_dev)``: rows of a third.. camera are folded into the first before the gather.
    ``drop_p = (attention's dropout, FFN hidden dropout, FFN output dropout)``: the active probabilities of train()
    mode (scale tensors are drawn here, in the reference's call order)."""
    idx, scale = gather
    for norm in (norm0, norm1):
        if not (isinstance(norm, torch.nn.LayerNorm) and tuple(norm.normalized_shape) == (256,) and norm.weight is not None
                and norm.bias is not None):
            return None
    if not (isinstance(fc1, torch.nn.Linear) and isinstance(fc2, torch.nn.Linear) and tuple(w0.shape) == (256, 256)
            and tuple(fc1.weight.shape) == (512, 256) and tuple(fc2.weight.shape) == (256, 512) and fc1.bias is not None
            and fc2.bias is not None and rows.shape[-1] == 256 and idx.dim() == 2 and idx.shape[1] == 2
            and idx.dtype == torch.int32 and row_slot.dtype == torch.int32 and res is not None
            and res.numel() == idx.shape[0] * 256 and rows.dim() == 2 and rows.data_ptr() % 16 == 0
            and wanted(rows, w0, b0, res, norm0.weight, norm0.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                       norm1.weight, norm1.bias)):
        return None
    drops = None
    if any(p > 0 for p in drop_p):
        M = idx.shape[0]
        lead = tuple(res.shape[:-1])
        drops = (dropout_scale(lead + (256,), drop_p[0], rows.device) if drop_p[0] > 0 else None,
                 dropout_scale(lead + (512,), drop_p[1], rows.device) if drop_p[1] > 0 else None,
                 dropout_scale(lead + (256,), drop_p[2], rows.device) if drop_p[2] > 0 else None, drop_p[1])
    return _SeamSFunction.apply(rows, w0, b0, res, norm0.weight, norm0.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                                norm1.weight, norm1.bias, idx, scale, row_slot, nrows, fold, norm0.eps, norm1.eps, tag, drops)
