"""The sampling operators (``msda``, ``msda_ragged``, the fused front end + sampling kernel with its three-step backward) and the
row kernels around them (residual + LayerNorm, camera mean)."""
import ctypes
import math
import os

import torch
from torch.autograd.function import Function, once_differentiable

from .. import _lib
from ..ext import _ptr, _req
from ..functions import MultiScaleDeformableAttnFunction_fp32

from .. import modes as _modes
from ._base import _NoTimer, _TIMER, _forward_modes, _m, _timed, _zero_scalar


def _pkg():
    """The ``ops`` package: calls between operators go through its namespace, so that a test which substitutes an
    operator there (``tests/helpers.py::oracle_ops``, the ``ops.linear`` spy of the camera-skipping test) sees them too."""
    import sys
    return sys.modules[__package__]



def msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
         im2col_step=64, tag="msda_fwd"):
    if _TIMER["cb"] is not None and not torch.is_grad_enabled():
        N, Q = sampling_locations.shape[:2]
        with _timed(tag, value, sampling_locations, attention_weights,
                    N * Q * value.shape[2] * value.shape[3]):
            return _msda(value, spatial_shapes, level_start_index, sampling_locations,
                         attention_weights, im2col_step)
    return _msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                 im2col_step)


def _msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
          im2col_step=64):
    if _m().value_storage == torch.bfloat16:
        from .functions import MultiScaleDeformableAttnFunction_bf16
        return MultiScaleDeformableAttnFunction_bf16.apply(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
            im2col_step).to(value.dtype)
    return MultiScaleDeformableAttnFunction_fp32.apply(
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
        im2col_step)


def _ragged_check(value, shapes, start, loc, attn, row_batch):
    _req(value.is_cuda, "bevmsda: value must be a GPU tensor (there is no CPU path)")
    dev = value.device
    for name, t in (("spatial_shapes", shapes), ("level_start_index", start),
                    ("sampling_locations", loc), ("attention_weights", attn),
                    ("row_batch", row_batch)):
        _req(t.device == dev and t.is_contiguous(), f"bevmsda: {name} must be contiguous on {dev}")
    _req(value.is_contiguous() and value.dim() == 4, "bevmsda: value must be contiguous (N,S,M,D)")
    _req(row_batch.dtype == torch.int32, "bevmsda: row_batch must be int32")
    _req(shapes.dtype == torch.int64 and start.dtype == torch.int64,
         "bevmsda: spatial_shapes / level_start_index must be int64")
    _req(loc.dtype == torch.float32 and attn.dtype == torch.float32,
         "bevmsda: sampling_locations / attention_weights must be float32")
    N, S, M, D = value.shape
    _req(loc.dim() == 5 and loc.shape[-1] == 2, "bevmsda: ragged sampling_locations must be (R,M,L,P,2)")
    R, _, L, P, _ = loc.shape
    _req(loc.shape[1] == M and tuple(attn.shape) == (R, M, L, P) and row_batch.numel() == R
         and tuple(shapes.shape) == (L, 2) and start.numel() == L,
         "bevmsda: inconsistent ragged operand shapes")
    return N, S, M, D, L, R, P


class _RaggedFunction(Function):

    @staticmethod
    def forward(ctx, value, shapes, start, loc, attn, row_batch, tag="msda_fwd"):
        ctx.modes = _m().snapshot()
        ctx.in_dtype = value.dtype
        store = _m().value_storage
        value = value.to(store).contiguous()
        loc = loc.float().contiguous()
        attn = attn.float().contiguous()
        N, S, M, D, L, R, P = _ragged_check(value, shapes, start, loc, attn, row_batch)
        lib = _lib.load()
        out = torch.empty((R, M * D), dtype=store, device=value.device)
        fn = lib.bevmsda_forward_ragged_f32 if store == torch.float32 else lib.bevmsda_forward_ragged_bf16
        with torch.cuda.device(value.device), _timed(tag, value, loc, attn, R * M * D):
            rc = fn(_ptr(value), _ptr(shapes), _ptr(start), _ptr(loc), _ptr(attn), _ptr(row_batch),
                    N, S, M, D, L, R, P, _ptr(out), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "msda_ragged forward")
        ctx.save_for_backward(value, shapes, start, loc, attn, row_batch)
        # rows go on to fp32 Linear layers: never hand the (bf16) STORAGE dtype downstream
        return out.to(torch.float32 if ctx.in_dtype == torch.bfloat16 else ctx.in_dtype)

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, grad_out):
        value, shapes, start, loc, attn, row_batch = ctx.saved_tensors
        N, S, M, D = value.shape
        R, _, L, P, _ = loc.shape
        grad_out = grad_out.to(value.dtype).contiguous()
        gv = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
        gl = torch.empty_like(loc)
        ga = torch.empty_like(attn)
        lib = _lib.load()
        fn = lib.bevmsda_backward_ragged_f32 if value.dtype == torch.float32 \
            else lib.bevmsda_backward_ragged_bf16
        with torch.cuda.device(value.device):
            rc = fn(_ptr(value), _ptr(shapes), _ptr(start), _ptr(loc), _ptr(attn), _ptr(row_batch),
                    _ptr(grad_out), N, S, M, D, L, R, P, _ptr(gv), _ptr(gl), _ptr(ga),
                    torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "msda_ragged backward")
        return gv.to(ctx.in_dtype), None, None, gl, ga, None, None


def msda_ragged(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                row_batch, tag="msda_fwd"):
    """value (N,S,M,D); sampling_locations (R,M,L,P,2); attention_weights
    (R,M,L,P); row_batch (R,) int32 in [0,N) -> (R, M*D)."""
    return _RaggedFunction.apply(value, spatial_shapes, level_start_index, sampling_locations,
                                 attention_weights, row_batch, tag)


def set_fused_front_end(flag):
    """Enable / disable the fused softmax + location + sampling kernel on the
    no-grad path (on by default; the autograd path always uses the unfused
    operator, whose backward kernels exist)."""
    _modes.process_defaults().fused = bool(flag)


def fused_wanted(*tensors):
    """The fused kernel is forward-only: use it when nothing asks for a gradient."""
    return _m().fused and not (torch.is_grad_enabled()
                                      and any(t is not None and t.requires_grad for t in tensors))


_RETIRED_FUSED_KWARGS = frozenset(("cam_start", "max_cam_rows", "lds_pixels"))


def msda_fused(value, spatial_shapes, level_start_index, proj, n_off, ref, row_batch, *, M, L, P,
               K, off_head, off_k, lg_head, lg_k, ref_mode, vmul, vadd, Q=0, row_src=None,
               tag="msda_fwd", nrows=None, launch_rows=0, save=None, grid_hw=None, **retired):
    """Sampling with the softmax / location prologue and the queue mean fused in
    (C ABI: ``bevmsda_fused_forward_*``, include/bevmsda.h).

    value (N,S,M,32); proj (R, C_out) raw output of the merged projection GEMM
    whose first ``n_off`` columns are sampling offsets and the rest attention
    logits; ref (R,K,A,2); row_batch (R,) int32 or None; row_src (R,) int32 or
    None: projection row read by output row r (proj then has one row per BEV
    query instead of one per output row).  Returns (R, M*32), or ``None`` when
    the shape is not covered (caller falls back to the unfused path).

    ``nrows`` ((1,) int32 device tensor, from a device-side frame plan): the ACTUAL number of
    rows; R above is then the capacity of the row arrays and the returned tensor has that many
    rows, of which only the first ``nrows`` are written (``bevmsda_fused_forward_rows_*``);
    ``launch_rows`` is the host's hint of that count (sizes the main launch; 0 = no hint).
    ``save = (loc (R, M, L, P, 2) or None, attn (R, M, L, P))`` fp32 (with ``nrows``; K = 1, P = 8, L >= 2): the kernel also
    writes the sampling locations and attention weights its rows used (``bevmsda_fused_forward_rows_save_*``) — what the
    operator's backward reads.  ``grid_hw = (height, width)``: the caller's HOST copy of the one level's shape when the rows
    are that grid's cells in raster order (TemporalSelfAttention over the BEV grid): lets the library take the kernel that stages
    a tile's tap lines in LDS (``modes.fused_spec = 5``, ``bevmsda_fused_desc.reserved[5]``)."""
    unknown = set(retired) - _RETIRED_FUSED_KWARGS
    if unknown:         # (the options of the retired LDS-staged kernels are still accepted and ignored; a typo is not)
        raise TypeError(f"msda_fused() got unexpected keyword arguments {sorted(unknown)}")
    _req(value.is_cuda, "bevmsda: value must be a GPU tensor (there is no CPU path)")
    store = _m().value_storage
    value = value.to(store)
    _req(value.is_contiguous() and value.dim() == 4, "bevmsda: value must be contiguous (N,S,M,D)")
    _req(proj.dtype == torch.float32 and proj.dim() == 2 and proj.stride(1) == 1,
         "bevmsda: proj must be a float32 (R, C) matrix with unit column stride")
    ref = ref.float().contiguous()
    N, S, Mv, D = value.shape
    R = proj.shape[0] if row_src is None else row_src.numel()
    A = ref.shape[-2]
    _req(Mv == M and ref.numel() == R * K * A * 2, "bevmsda: inconsistent fused operand shapes")
    if row_src is not None:
        _req(row_src.dtype == torch.int32 and row_src.is_contiguous() and row_src.device == proj.device,
             "bevmsda: row_src must be a contiguous int32 (R,) tensor")
    if row_batch is not None:
        _req(row_batch.dtype == torch.int32 and row_batch.numel() == R and row_batch.is_contiguous(),
             "bevmsda: row_batch must be a contiguous int32 (R,) tensor")
    desc = _lib.FusedDesc(R=R, proj_row=proj.stride(0), N=N, S=S, M=M, D=D, L=L, P=P, Q=Q, K=K, A=A,
                          ref_mode=ref_mode, off_head=off_head, off_k=off_k, lg_head=lg_head,
                          lg_k=lg_k, vmul=vmul, vadd=vadd)
    if _m().fused_spec:                     # A/B knob of the specialised bodies (modes.Modes.fused_spec)
        desc.reserved[5] = _m().fused_spec
    if grid_hw is not None and nrows is None and L == 1 and 0 < grid_hw[0] < 32768 and 0 < grid_hw[1] < 65536 \
            and grid_hw[0] * grid_hw[1] == R:
        desc.reserved[3] = (int(grid_hw[0]) << 16) | int(grid_hw[1])
    if _m().fused_wpe and nrows is None:   # benchmark sweeps: register budget of the kernel
        desc.reserved[0] = _m().fused_wpe
    if _m().fused_lds_pad_kb and L > 1:    # occupancy cap of the multi-level (SCA) launch
        desc.reserved[4] = _m().fused_lds_pad_kb
    lib = _lib.load()
    if store == torch.bfloat16 and not _m().bf16_lanes8:
        desc.reserved[2] = 1            # 16-byte-lane kernel writes fp32 rows for the fp32 output projection
        out = torch.empty((R, M * D), dtype=torch.float32, device=value.device)
    else:
        if store == torch.bfloat16:
            desc.reserved[1] = 1        # benchmark knob: the 8-byte-lane bf16 kernel
        out = torch.empty((R, M * D), dtype=store, device=value.device)
    fn = lib.bevmsda_fused_forward_f32 if store == torch.float32 else lib.bevmsda_fused_forward_bf16
    logits = proj[:, n_off:]
    with torch.cuda.device(value.device):
        # algorithmic bytes: value + raw projection row (offsets 8 B + logit 4 B per point) + out
        alg = value.numel() * value.element_size() + R * M * K * L * P * 12 \
            + R * M * D * out.element_size()
        if nrows is not None:       # the row count is on the device: (fixed bytes, bytes per row), resolved by the hook's owner
            alg = ("per_row", value.numel() * value.element_size(), M * K * L * P * 12 + M * D * out.element_size())
        cb = _TIMER["cb"]
        ctx = cb(tag, alg) if cb is not None else _NoTimer()
        with ctx:
            if nrows is not None:
                _req(nrows.dtype == torch.int32 and nrows.is_cuda and nrows.numel() >= 1,
                     "bevmsda: nrows must be an int32 device tensor")
                hint = int(max(0, min(launch_rows, R)))
                cap_launch = _m().fused_capacity_launch
                if cap_launch == "auto":
                    cap_launch = 0 < hint and R - hint <= FUSED_CAPACITY_AUTO_ROWS
                desc.reserved[3] = R if cap_launch else hint
                extra = ()
                if save is not None:
                    sl, sa = save
                    _req(sa.dtype == torch.float32 and sa.is_contiguous() and sa.numel() == R * M * L * P
                         and (sl is None or (sl.dtype == torch.float32 and sl.is_contiguous() and sl.numel() == R * M * L * P * 2)),
                         "bevmsda: save = (loc (R, M, L, P, 2) or None, attn (R, M, L, P)) contiguous fp32 tensors")
                    fnr = lib.bevmsda_fused_forward_rows_save_f32 if store == torch.float32 \
                        else lib.bevmsda_fused_forward_rows_save_bf16
                    extra = (_ptr(sl) if sl is not None else None, _ptr(sa))
                else:
                    fnr = lib.bevmsda_fused_forward_rows_f32 if store == torch.float32 \
                        else lib.bevmsda_fused_forward_rows_bf16
                rc = fnr(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), proj.data_ptr(),
                         logits.data_ptr(), _ptr(ref), _ptr(row_batch) if row_batch is not None else None,
                         _ptr(row_src) if row_src is not None else None, nrows.data_ptr(),
                         ctypes.byref(desc), _ptr(out), *extra, torch.cuda.current_stream().cuda_stream)
            else:
                _req(save is None, "bevmsda: save needs the device-side row count form (nrows)")
                rc = fn(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), proj.data_ptr(),
                        logits.data_ptr(), _ptr(ref), _ptr(row_batch) if row_batch is not None else None,
                        _ptr(row_src) if row_src is not None else None, ctypes.byref(desc), _ptr(out), torch.cuda.current_stream().cuda_stream)
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "msda_fused forward")
    return out




def set_fused_training(flag):
    """Autograd path of the attention modules through the fused sampling kernel + its three-step backward
    (default) or through the unfused operator with the softmax / location arithmetic as torch ops."""
    _modes.process_defaults().fused_train = bool(flag)


def fused_training_wanted(*tensors):
    return _m().fused and _m().fused_train and torch.is_grad_enabled() and all(t is None or t.is_cuda for t in tensors) \
        and any(t is not None and t.requires_grad for t in tensors) and not torch.is_autocast_enabled()


_DEBUG = {"tap": None}        # forensics hook (tools/ddp_diag.py): called with every operand and result of a fused sampling backward


class _FusedSampleFunction(Function):
    """``msda_fused`` under autograd.  Forward: the fused kernel (softmax, locations, sampling, queue mean from
    the raw projection rows) — nothing but its inputs is saved.  Backward (include/bevmsda.h,
    ``bevmsda_frontend_*``): (1) locations / weights / value batch entries of every (row, queue entry)
    recomputed from the projection rows, (2) the operator's backward kernels, (3) softmax backward and
    1 / (W, H) back onto the projection rows, accumulated over the rows that share one."""

    @staticmethod
    def forward(ctx, value, proj, shapes, start, ref, row_batch, row_src, n_off, meta, tag, q_rows=None, nrows=None,
                launch_rows=0, value_sink=None, n_extra=None):
        ctx.n_extra = n_extra             # (1,) int32 device tensor: the plan's count of slots with more than two rows
        ctx.modes = _m().snapshot()
        # (sink, i): the fp32 grad_value of this call is DEPOSITED in sink[i] and a zero-stride placeholder of the value's
        # own (bf16) dtype goes back through autograd — the producer of a bf16-stored value (train_ops.grouped_linear)
        # reads the sink, so the gradient is neither rounded to bf16 nor copied on its way there
        ctx.value_sink = value_sink
        # bf16 storage: ONE rounded copy of the value serves the forward kernel and, saved, the backward kernels
        vs = value.detach().to(_m().value_storage).contiguous()
        dyn = {} if nrows is None else dict(nrows=nrows, launch_rows=launch_rows)
        # SCA's shape on the device-side row count: the forward kernel writes the locations / weights of its rows, the
        # backward reads them instead of recomputing them (step (1) of the docstring: 79 us per base layer)
        saved = None
        if nrows is not None and meta["K"] == 1 and meta["P"] == 8 and meta["L"] >= 2 and row_src is not None \
                and meta["vmul"] == 1 and meta["vadd"] == 0 and row_batch is not None and _m().fused_save \
                and not (_m().value_storage == torch.bfloat16 and _m().bf16_lanes8):
            Rr, Mh, Lv, Pp = row_src.numel(), meta["M"], meta["L"], meta["P"]
            # fused_save = 2: the weights only — 4 of the 12 bytes per sampling point; the backward kernels recompute the locations
            # from proj / ref with the forward's own expression (bevmsda_backward_rows_offs_*; pillar-anchor references only)
            lean = _m().fused_save == 2 and meta["ref_mode"] == 0 and proj.stride(0) % 2 == 0 and meta["off_head"] % 2 == 0 \
                and proj.data_ptr() % 8 == 0
            saved = (None if lean else torch.empty((Rr, Mh, Lv, Pp, 2), dtype=torch.float32, device=vs.device),
                     torch.empty((Rr, Mh, Lv, Pp), dtype=torch.float32, device=vs.device))
            dyn["save"] = saved
            saved = tuple(t for t in saved if t is not None)
        out = _pkg().msda_fused(vs, shapes, start, proj.detach(), n_off, ref, row_batch, row_src=row_src,
                         tag=tag, **meta, **dyn)
        if out is None:
            raise RuntimeError("bevmsda: fused sampling kernel does not cover this call")
        ctx.value_dtype = value.dtype
        ctx.has_saved = saved is not None
        ctx.save_for_backward(vs, proj, shapes, start, ref.float().contiguous(),
                              row_batch if row_batch is not None else shapes.new_empty(0),
                              row_src if row_src is not None else shapes.new_empty(0),
                              *(saved if saved is not None else ()))
        ctx.n_off, ctx.meta, ctx.tag = n_off, meta, tag
        ctx.store = _m().value_storage     # the value storage the forward sampled (bf16: rounded copy of `value`)
        ctx.q_rows = q_rows               # (slots, J) int32 rows of every projection row, or None
        ctx.nrows = nrows                 # (1,) int32 device tensor: the row arrays above have CAPACITY rows
        return out

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, grad_out):
        value, proj, shapes, start, ref, row_batch, row_src = ctx.saved_tensors[:7]
        kept = ctx.saved_tensors[7:] if ctx.has_saved else None
        row_batch = row_batch if row_batch.numel() else None
        row_src = row_src if row_src.numel() else None
        m = ctx.meta
        M, L, P, K = m["M"], m["L"], m["P"], m["K"]
        N, S, _, D = value.shape
        R = proj.shape[0] if row_src is None else row_src.numel()
        A = ref.shape[-2]
        desc = _lib.FusedDesc(R=R, proj_row=proj.stride(0), N=N, S=S, M=M, D=D, L=L, P=P, Q=m.get("Q", 0), K=K, A=A,
                              ref_mode=m["ref_mode"], off_head=m["off_head"], off_k=m["off_k"],
                              lg_head=m["lg_head"], lg_k=m["lg_k"], vmul=m["vmul"], vadd=m["vadd"])
        dev = value.device
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        RK = R * K
        lean = kept is not None and len(kept) == 1          # the forward kept its weights only: locations from proj / ref
        if lean:
            loc, attn, rbk = None, kept[0], row_batch
        elif kept is not None:
            loc, attn, rbk = kept[0], kept[1], row_batch        # (K = 1: the value batch entry of a row is its row_batch)
        else:
            loc = torch.empty((RK, M, L, P, 2), dtype=torch.float32, device=dev)
            attn = torch.empty((RK, M, L, P), dtype=torch.float32, device=dev)
            rbk = torch.empty(RK, dtype=torch.int32, device=dev)
        bf = ctx.store == torch.bfloat16
        value = value.detach().to(ctx.store).contiguous()      # bf16 storage: the rounded values the forward saw
        proj = proj.detach()
        logits = proj[:, ctx.n_off:]
        nrows = ctx.nrows
        if nrows is not None and (K != 1 or row_batch is None or row_src is None or ctx.q_rows is None):
            raise RuntimeError("bevmsda: a device-side row count needs the ragged single-entry form with a q_rows table")
        with torch.cuda.device(dev):
            if kept is not None:
                pass                        # (the forward kernel wrote them: bevmsda_fused_forward_rows_save_*)
            elif nrows is not None:
                _lib.check(lib.bevmsda_frontend_expand_rows_f32(
                    proj.data_ptr(), logits.data_ptr(), _ptr(ref), _ptr(row_batch), _ptr(row_src), nrows.data_ptr(),
                    _ptr(shapes), ctypes.byref(desc), _ptr(loc), _ptr(attn), _ptr(rbk), st), "fused backward: expand (rows)")
            else:
                _lib.check(lib.bevmsda_frontend_expand_f32(
                    proj.data_ptr(), logits.data_ptr(), _ptr(ref), _ptr(row_batch) if row_batch is not None else None,
                    _ptr(row_src) if row_src is not None else None, _ptr(shapes), ctypes.byref(desc), _ptr(loc),
                    _ptr(attn), _ptr(rbk), st), "fused backward: expand")
            # queue entries that share their output row (TSA's mean over K entries, one value batch entry each, one batch
            # element): the backward kernels read scale * grad_out[r % R] themselves — no scaled, repeated copy
            shared = row_batch is None and K > 1 and N == K and m["vmul"] == K and m["vadd"] == 1 and m.get("Q", 0) == R \
                and nrows is None and grad_out.dim() == 2 and grad_out.shape[0] == R
            if bf and grad_out.dtype == torch.float32 and grad_out.is_contiguous() and grad_out.dim() == 2 \
                    and grad_out.shape[0] == R and grad_out.shape[1] % 8 == 0 and grad_out.data_ptr() % 16 == 0:
                # bf16 storage: round (and, without row sharing, scale by 1 / K per queue entry) over the rows that exist
                reps = 1 if shared else K
                g = torch.empty((R * reps, grad_out.shape[1]), dtype=torch.bfloat16, device=dev)
                for k in range(reps):
                    _lib.check(lib.bevmsda_cast_rows_bf16(
                        _ptr(grad_out), nrows.data_ptr() if nrows is not None else None, R, grad_out.shape[1],
                        1.0 if shared else 1.0 / K, g.data_ptr() + k * R * grad_out.shape[1] * 2, st), "fused backward: cast rows")
            elif shared:
                g = grad_out.to(ctx.store).contiguous()
            else:
                g = grad_out.float()
                if K > 1:       # out = mean over the queue entries; rows are queue-major
                    g = (g * (1.0 / K)).repeat(K, 1)
                g = g.to(ctx.store).contiguous()
            # grad_value: a dense zero-filled array, or (a sink with an arena) this call's pixel rows of the frame-wide
            # array that holds the value gradients of all layers side by side
            gv, gvs = None, 0
            if ctx.value_sink is not None and (nrows is not None or shared) and hasattr(ctx.value_sink[0], "buffer"):
                gv = ctx.value_sink[0].buffer(ctx.value_sink[1], value.shape, dev)
                gvs = 0 if gv is None else gv.stride(1)
            if gv is None:
                gv = torch.zeros(value.shape, dtype=torch.float32, device=dev)
            gl = torch.empty((RK, M, L, P, 2), dtype=torch.float32, device=dev)
            ga = torch.empty_like(attn)
            # algorithmic bytes of the operator's backward: value + locations + weights + grad_out read,
            # grad_value + grad_loc + grad_attn written
            cb = _TIMER["cb"]
            alg = value.numel() * value.element_size() + RK * M * L * P * 12 + RK * M * D * g.element_size() \
                + value.numel() * 4 + RK * M * L * P * 12
            if nrows is not None:       # (fixed bytes, bytes per row): resolved by the hook's owner from the frame's row count
                alg = ("per_row", value.numel() * value.element_size() + value.numel() * 4,
                       M * L * P * 24 + M * D * g.element_size())
            with (cb(ctx.tag.replace("_fwd", "") + "_bwd", alg) if cb is not None else _NoTimer()):
                if lean:
                    src = _lib.LocSource(offs=proj.data_ptr(), ref=_ptr(ref), row_src=_ptr(row_src), proj_row=proj.stride(0),
                                         off_head=m["off_head"], A=A)
                    _lib.check((lib.bevmsda_backward_rows_offs_bf16 if bf else lib.bevmsda_backward_rows_offs_f32)(
                        _ptr(value), _ptr(shapes), _ptr(start), ctypes.byref(src), _ptr(attn), _ptr(rbk), _ptr(g), nrows.data_ptr(),
                        N, S, M, D, L, RK, P, _ptr(gv), gvs, _ptr(gl), _ptr(ga), st), "fused backward: operator (rows, locations recomputed)")
                elif nrows is not None:
                    _lib.check((lib.bevmsda_backward_rows_bf16 if bf else lib.bevmsda_backward_rows_f32)(
                        _ptr(value), _ptr(shapes), _ptr(start), _ptr(loc), _ptr(attn), _ptr(rbk), _ptr(g), nrows.data_ptr(),
                        N, S, M, D, L, RK, P, _ptr(gv), gvs, _ptr(gl), _ptr(ga), st), "fused backward: operator (rows)")
                elif shared:
                    # one batch element, one value batch entry per queue entry: queue-major rows ARE the dense
                    # (N = K, Q = R) layout of the operator (its grid-tiled grad_value path applies)
                    rc = (lib.bevmsda_backward_shared_bf16 if bf else lib.bevmsda_backward_shared_f32)(
                        _ptr(value), _ptr(shapes), _ptr(start), _ptr(loc), _ptr(attn), _ptr(g), R, 1.0 / K, N, S, M, D, L, R, P,
                        _ptr(gv), gvs, _ptr(gl), _ptr(ga), st)
                    if rc == _lib.ERR_UNSUPPORTED:      # (a shape only the first-generation kernels take: scaled, repeated rows)
                        if gvs:
                            gv, gvs = torch.zeros(value.shape, dtype=torch.float32, device=dev), 0
                        g = (g.float() * (1.0 / K)).repeat(K, 1).to(ctx.store).contiguous()
                        rc = (lib.bevmsda_backward_bf16 if bf else lib.bevmsda_backward_f32)(
                            _ptr(value), _ptr(shapes), _ptr(start), _ptr(loc), _ptr(attn), _ptr(g), N, S, M, D, L, R, P,
                            _ptr(gv), _ptr(gl), _ptr(ga), st)
                    _lib.check(rc, "fused backward: operator")
                else:
                    _lib.check((lib.bevmsda_backward_ragged_bf16 if bf else lib.bevmsda_backward_ragged_f32)(
                        _ptr(value), _ptr(shapes), _ptr(start), _ptr(loc), _ptr(attn), _ptr(rbk), _ptr(g), N, S, M, D, L,
                        RK, P, _ptr(gv), _ptr(gl), _ptr(ga), st), "fused backward: operator")
            qr = ctx.q_rows
            rc = _lib.ERR_UNSUPPORTED
            if row_src is not None and qr is not None and K == 1 and qr.shape[0] == proj.shape[0]:
                gproj = torch.empty_like(proj)
                rc = lib.bevmsda_frontend_chain_gather_f32(
                    _ptr(gl), _ptr(ga), _ptr(attn), _ptr(qr), qr.shape[0], qr.shape[1],
                    ctx.n_extra.data_ptr() if ctx.n_extra is not None else None, _ptr(shapes),
                    ctypes.byref(desc), gproj.data_ptr(), gproj[:, ctx.n_off:].data_ptr(), st)
                if rc != _lib.ERR_UNSUPPORTED:
                    _lib.check(rc, "fused backward: chain (gather)")
            if rc == _lib.ERR_UNSUPPORTED and nrows is not None:
                # the atomic chain pass below has no row count: it would walk all CAPACITY rows and add the uninitialised
                # gl / ga rows beyond the count into gproj (the callers only send level counts the gather kernel takes)
                raise RuntimeError("bevmsda: a device-side row count needs bevmsda_frontend_chain_gather_f32 (L in 1, 2, 4)")
            if rc == _lib.ERR_UNSUPPORTED:
                # (with row_src the chain pass ADDS the rows of a query with atomics; without it every element is stored)
                gproj = torch.zeros_like(proj) if row_src is not None else torch.empty_like(proj)
                _lib.check(lib.bevmsda_frontend_chain_f32(
                    _ptr(gl), _ptr(ga), _ptr(attn), _ptr(row_src) if row_src is not None else None, _ptr(shapes),
                    ctypes.byref(desc), gproj.data_ptr(), gproj[:, ctx.n_off:].data_ptr(), st), "fused backward: chain")
        if _DEBUG["tap"] is not None:
            _DEBUG["tap"](ctx.tag, dict(proj=proj, ref=ref, loc=loc, attn=attn, g=g, gl=gl, ga=ga, gproj=gproj, nrows=nrows))
        if ctx.value_sink is not None:
            sink, slot = ctx.value_sink
            sink[slot] = gv
            gv_out = _zero_scalar(ctx.value_dtype, dev).expand(value.shape)
        else:
            gv_out = gv.to(ctx.value_dtype)
        return gv_out, gproj, None, None, None, None, None, None, None, None, None, None, None, None, None


def msda_fused_autograd(value, spatial_shapes, level_start_index, proj, n_off, ref, row_batch, *, row_src=None,
                        q_rows=None, tag="msda_fwd", nrows=None, launch_rows=0, value_sink=None, n_extra=None, **meta):
    """``msda_fused`` with gradients w.r.t. ``value`` and ``proj`` (D = 32; fp32 or bf16 value storage — with bf16
    the forward's rounded copy of ``value`` is what the backward kernels read; the caller checks
    ``fused_training_wanted``).  ``proj`` must be the projection matrix itself (offsets in the first ``n_off``
    columns, logits behind them, every column of a row used by exactly one (head, queue entry, level, point)).
    ``q_rows`` (slots, J): the rows that read each projection row (inverse of ``row_src``) — with it the
    backward's last step is a gather (stores) instead of atomics."""
    if q_rows is not None:
        _req(q_rows.dtype == torch.int32 and q_rows.dim() == 2 and q_rows.is_contiguous() and q_rows.device == proj.device,
             "bevmsda: q_rows must be a contiguous int32 (slots, J) device tensor")
    return _FusedSampleFunction.apply(value, proj, spatial_shapes, level_start_index, ref, row_batch, row_src, n_off,
                                      meta, tag, q_rows, nrows, launch_rows, value_sink, n_extra)


FUSED_CAPACITY_AUTO_ROWS = 65536       # (modes.fused_capacity_launch = "auto": surplus rows up to which ONE capacity-sized launch is used)


def fold_extra_rows(rows, q_rows_all, n_extra):
    """In place: rows[q_rows_all[s, 0]] += sum_{j >= 2} rows[q_rows_all[s, j]] for the slots more
    than two cameras see (``bevmsda_fold_extra_rows_f32``; a no-op launch when the device counter
    ``n_extra`` is 0), so that the two-row gather of ``linear_gather_mean`` covers every camera."""
    _req(rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2 and rows.stride(1) == 1,
         "bevmsda: rows must be a float32 (R, C) GPU matrix")
    J = q_rows_all.shape[1]
    if J <= 2:
        return rows
    lib = _lib.load()
    with torch.cuda.device(rows.device):
        rc = lib.bevmsda_fold_extra_rows_f32(_ptr(rows), rows.stride(0), _ptr(q_rows_all), q_rows_all.shape[0],
                                             J, rows.shape[1], n_extra.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "fold_extra_rows")
    return rows


def add_layernorm(x, res, weight, bias, eps):
    """LayerNorm(x + res) over the last dim in one pass (C ABI:
    ``bevmsda_add_layernorm_f32``); returns ``None`` when the shape / dtype is
    not covered so that the caller runs the separate torch ops."""
    C = x.shape[-1]
    if not (x.is_cuda and x.dtype == torch.float32 and C in (256, 512, 1024)
            and weight is not None and bias is not None and weight.dtype == torch.float32):
        return None
    x = x.contiguous()
    res = res.contiguous() if res is not None else None
    if res is not None and (res.shape != x.shape or res.dtype != x.dtype):
        return None
    out = torch.empty_like(x)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.bevmsda_add_layernorm_f32(_ptr(x), _ptr(res) if res is not None else None,
                                           _ptr(weight), _ptr(bias), float(eps), x.numel() // C, C,
                                           _ptr(out), torch.cuda.current_stream().cuda_stream)
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "add_layernorm")
    return out


class _AddLayerNormFunction(Function):
    """``LayerNorm(x + res)`` under autograd on the row kernels: forward = ``add_layernorm`` (one pass), backward =
    ``bevmsda_add_layernorm_backward_f32`` (x + res recomputed; one gradient for both addends, column sums for
    weight / bias).  Replaces torch's add + LayerNorm forward (2 passes) and its 3-kernel backward."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        ctx.modes = _m().snapshot()
        x = x.contiguous()
        res = res.contiguous()
        y = add_layernorm(x.detach(), res.detach(), weight.detach(), bias.detach(), eps)
        if y is None:
            raise RuntimeError("bevmsda: add_layernorm does not cover this call")
        ctx.save_for_backward(x, res, weight)
        ctx.eps = float(eps)
        return y

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, g):
        x, res, weight = ctx.saved_tensors
        C = x.shape[-1]
        g = g.float().contiguous()
        gx = torch.empty_like(x)
        lib = _lib.load()
        rows = x.numel() // C
        parts = int(lib.bevmsda_add_layernorm_backward_partials(rows))
        scratch = torch.empty(max(parts, 1) * 2 * C, dtype=torch.float32, device=x.device)
        gwb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.bevmsda_add_layernorm_backward_f32(
                _ptr(x), _ptr(res), _ptr(weight), _ptr(g), ctx.eps, rows, C, _ptr(gx), _ptr(scratch),
                _ptr(gwb), torch.cuda.current_stream().cuda_stream), "add_layernorm backward")
        return gx, gx, gwb[0], gwb[1], None


def add_layernorm_autograd(x, res, norm):
    """``norm(x + res)`` with gradients, or ``None`` when the kernels do not cover the call."""
    C = x.shape[-1]
    if not (_m().fused_train and x.is_cuda and x.dtype == torch.float32 and res.dtype == torch.float32
            and res.shape == x.shape and C in (256, 512) and isinstance(norm, torch.nn.LayerNorm)
            and norm.weight is not None and norm.bias is not None and norm.weight.dtype == torch.float32
            and tuple(norm.normalized_shape) == (C,) and not torch.is_autocast_enabled()):
        return None
    return _AddLayerNormFunction.apply(x, res, norm.weight, norm.bias, norm.eps)


def gather_mean(rows, idx, scale):
    """out[q] = scale[q] * sum_j rows[idx[q, j]] (idx int32, -1 = empty): the SCA
    scatter-add + camera-count division as a gather (``bevmsda_gather_mean_f32``)."""
    _req(rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2 and rows.is_contiguous(),
         "bevmsda: rows must be a contiguous float32 (R, C) GPU tensor")
    _req(idx.dtype == torch.int32 and idx.dim() == 2 and idx.is_contiguous(),
         "bevmsda: idx must be a contiguous int32 (Q, J) tensor")
    Qn, J = idx.shape
    scale = scale.reshape(-1).float().contiguous()
    _req(scale.numel() == Qn, "bevmsda: scale must have one entry per output row")
    C = rows.shape[1]
    out = torch.empty((Qn, C), dtype=torch.float32, device=rows.device)
    lib = _lib.load()
    with torch.cuda.device(rows.device):
        rc = lib.bevmsda_gather_mean_f32(_ptr(rows), _ptr(idx), _ptr(scale), Qn, J, C, _ptr(out),
                                         torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "gather_mean")
    return out


class _GatherMeanFunction(Function):
    """``gather_mean`` under autograd: SpatialCrossAttention's per-camera scatter-add + camera-count division
    (spatial_cross_attention.py:165-172) as one gather kernel forward; backward: every row takes its slot's
    gradient times the slot's scale (a gather by ``row_slot``)."""

    @staticmethod
    def forward(ctx, rows, idx, scale, row_slot):
        ctx.modes = _m().snapshot()
        ctx.save_for_backward(scale.reshape(-1).float(), row_slot)
        return _pkg().gather_mean(rows.detach().contiguous(), idx, scale)

    @staticmethod
    @once_differentiable
    @_forward_modes
    def backward(ctx, g):
        scale, row_slot = ctx.saved_tensors
        return (g * scale[:, None]).index_select(0, row_slot), None, None, None


def gather_mean_autograd(rows, idx, scale, row_slot):
    """``gather_mean(rows, idx, scale)`` with a gradient w.r.t. ``rows``; ``row_slot`` (R,) int64 = the slot every
    row belongs to (``idx`` inverted: the frame plan's ``row_query``)."""
    return _GatherMeanFunction.apply(rows, idx, scale, row_slot)
