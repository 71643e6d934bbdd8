"""The dense projections: MFMA kernels behind ``linear`` and its fused forms, weight images, ``KERNEL_SELECTION``, the autograd Function."""
import ctypes
import math
import os

import torch
from torch.autograd.function import Function, once_differentiable

from .. import _lib
from ..ext import _ptr, _req
from ..functions import MultiScaleDeformableAttnFunction_fp32

from .. import modes as _modes
from ._base import _NoTimer, _forward_modes, _m, _ver
from .sampling import fused_wanted


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


def _cache_ok(weight):
    """Derived images of a weight (packed / transposed copies) are cached per version — except while a HIP graph of a
    TRAINING step is being captured over a trainable weight (grad mode on): the replayed graph must rebuild them from
    the weight's current values (an optimizer step between replays changes them without the capture noticing), so the
    conversion kernels are captured too.  A graph captured under ``torch.no_grad()`` is an inference graph: it freezes
    the images it was captured with, exactly as it freezes the merged (concatenated) projection weights — change the
    weights, capture again (round 5: the 24 re-packing launches per replayed forward step were 2.8 % of it)."""
    if not weight.is_cuda or not torch.cuda.is_current_stream_capturing():
        return True
    if weight.requires_grad and (torch.is_grad_enabled() or _m().graph_repack):
        return False
    # Round 6: the autograd Functions run their forward and backward with grad mode OFF, so the rule above never saw the
    # captures it was written for — a captured TRAINING step froze the images of every weight that reached the kernels
    # as the parameter object itself (the FFN's: output off by 4e-3, gradients by 7 % after one optimizer step between
    # replays; tools/probes/graph_update_check.py).  What tells a training capture is the step state: between
    # ``begin_training_step(module)`` and the next forward without gradients, images of memory that belongs to the module's
    # parameters are rebuilt inside the capture (by the one-launch rebuild when it covers them, else by their own launch).
    if _TRAIN["active"]:
        try:
            return weight.untyped_storage().data_ptr() not in _TRAIN["storages"]
        except Exception:       # noqa: BLE001
            return False
    return True


# Weight images a stream capture handed to a graph, by id(image): {image (kept alive: the graph holds its address), a weak
# reference to the weight it was made from, the weight's version and address at capture}.  A graph captured under
# torch.no_grad() FREEZES these images (and the merged projection weights): an in-place weight update between replays
# (load_state_dict, EMA, an optimizer step between periodic graphed evaluations) replays the old weights —
# ``graph_weights_stale()`` / ``assert_graph_weights_fresh()`` say so, ``release_captured_images()`` drops the registry
# once the graphs are gone.
_CAPTURED_IMAGES = {}


def _cached_image(hit, weight=None):
    if torch.cuda.is_current_stream_capturing() and id(hit[1]) not in _CAPTURED_IMAGES:
        import weakref
        try:
            ref = weakref.ref(weight) if weight is not None else None
        except TypeError:
            ref = None
        _CAPTURED_IMAGES[id(hit[1])] = dict(image=hit[1], weight=ref, version=_ver(weight) if weight is not None else None,
                                            data_ptr=weight.data_ptr() if weight is not None else None,
                                            shape=tuple(weight.shape) if weight is not None else None)
    return hit[1]


def graph_weights_stale():
    """Weights whose packed / panel / transposed images were frozen into a captured HIP graph and that have been written
    to (or moved, or freed) since: list of ``(shape, reason)``.  Empty = every captured graph still replays current weights."""
    out = []
    for rec in _CAPTURED_IMAGES.values():
        if rec["weight"] is None:
            continue
        w = rec["weight"]()
        if w is None:
            out.append((rec["shape"], "the weight tensor was freed"))
        elif w.data_ptr() != rec["data_ptr"]:
            out.append((rec["shape"], "the weight tensor was moved / reallocated"))
        elif _ver(w) != rec["version"]:
            out.append((rec["shape"], "the weight was written to after the capture"))
    return out


def assert_graph_weights_fresh():
    """Raise when a captured inference graph would replay weights that have changed since its capture (call before
    ``graph.replay()`` wherever weights can change between replays; re-capture to pick the new values up)."""
    stale = graph_weights_stale()
    if stale:
        raise RuntimeError("bevmsda: %d weight image(s) frozen into a captured HIP graph are stale (%s ...): the weights "
                           "changed after the capture — capture the graph again" % (len(stale), stale[:3]))


def release_captured_images():
    """Forget the images captured graphs hold (call after destroying those graphs: the registry keeps the images alive)."""
    n = len(_CAPTURED_IMAGES)
    _CAPTURED_IMAGES.clear()
    return n


# ---------------------------------------------------------------------------------------------------------------------
# Weight images of a TRAINING step (round 6).  The weights change between steps, so a training step rebuilds every image
# it uses from their current values — until round 5 image by image, 52 launches of ~5 us each per step at base (also
# inside the captured graph of a step).  Now: an image packed under grad mode from memory that belongs to a parameter of
# the module whose step is running is REGISTERED (its blob keeps its address), and ``begin_training_step(module)`` — the
# encoder calls it when a differentiable forward starts — rebuilds all registered images with ONE launch
# (``bevmsda_linear_pack_weights_multi_f32``) and marks them fresh for this step; ``packed_weight`` / ``panel_weight``
# then hand out the fresh blob without a launch.  Safety net for calls outside a step: an image is only handed out while
# the version counter of the tensor it was packed from is the one recorded at the rebuild.
# (``active``: between the start of a differentiable forward and the next forward without gradients — the autograd
# Functions run their forward AND backward with grad mode off, so grad mode cannot tell a training step from inference)
_TRAIN = {"step": 0, "entries": {}, "table": None, "table_key": None, "table_blocks": 0, "storages": frozenset(),
          "multi_launches": 0, "single_launches": 0, "enabled": __import__("os").environ.get("BEVMSDA_IMAGE_BATCH", "1") == "1",
          "active": False}


def set_training_image_batching(flag):
    """A/B switch of the one-launch image rebuild (default on)."""
    _TRAIN["enabled"] = bool(flag)
    _TRAIN["active"] = False
    _TRAIN["entries"].clear()
    _TRAIN["table"] = _TRAIN["table_key"] = None


def training_image_stats():
    return dict(step=_TRAIN["step"], images=len(_TRAIN["entries"]), multi_launches=_TRAIN["multi_launches"],
                single_launches=_TRAIN["single_launches"], unregistered=dict(_TRAIN.get("unregistered", {})))


def _train_key(kind, weight):
    return (kind, _is_transposed_view(weight), weight.data_ptr(), tuple(weight.shape), weight.stride(0), weight.stride(1))


def _train_image(kind, weight):
    """The fresh image of ``weight`` rebuilt at the start of this training step, or None."""
    if not (_TRAIN["enabled"] and _TRAIN["active"] and _TRAIN["entries"]):
        return None
    e = _TRAIN["entries"].get(_train_key(kind, weight))
    if e is None or e["fresh_step"] != _TRAIN["step"] or e["version"] != _ver(weight):
        return None
    e["used_step"] = _TRAIN["step"]
    return e["blob"]


def _train_register(kind, weight, blob, launched=True):
    """Called after an image was packed the single way (or found in the per-version cache) during a training step: from the
    next step on it is rebuilt in the batch."""
    _TRAIN["single_launches"] += int(_TRAIN["active"] and launched)
    if not (_TRAIN["enabled"] and _TRAIN["active"] and weight.is_cuda):
        return
    if torch.cuda.is_current_stream_capturing():
        return              # (a blob allocated in a capture's private pool is that graph's: never adopted)
    try:
        sp = weight.untyped_storage().data_ptr()
    except Exception:       # noqa: BLE001
        return
    if sp not in _TRAIN["storages"]:
        # (not a parameter of the module whose step is running: a derived tensor may move or die)
        _TRAIN.setdefault("unregistered", {})[(kind, tuple(weight.shape))] = "memory outside the module's parameters"
        return
    N, K = weight.shape
    t = _is_transposed_view(weight)
    _TRAIN["entries"][_train_key(kind, weight)] = dict(
        # (a DETACHED alias: same memory, same version counter — the weight itself may carry a grad_fn, and a reference to
        # it would keep that step's autograd graph, with the streams its nodes were created on, alive into the next steps:
        # a later graph capture then ran those nodes' gradient accumulation on the old stream and hipStreamEndCapture crashed)
        blob=blob, tensor=weight.detach(), storage=sp, N=N, K=K, ldw=weight.stride(1) if t else weight.stride(0),
        kind=(1 if t else 0) | (2 if kind == "panel" else 0), used_step=_TRAIN["step"], fresh_step=_TRAIN["step"],
        version=_ver(weight))


def begin_training_step(module):
    """Start of a differentiable forward of ``module``: ONE launch rebuilds every weight image registered in the previous
    steps from the weights' current values (nothing on the first step: images register as they are packed)."""
    st = _TRAIN
    st["active"] = True
    if not torch.cuda.is_current_stream_capturing() or st.get("storages_of") != id(module):
        st["storages"] = frozenset(p.untyped_storage().data_ptr() for p in module.parameters() if p.is_cuda)
        st["storages_of"] = id(module)
    if not st["enabled"]:
        return
    st["step"] += 1
    # images not used for two steps, or whose memory no longer belongs to this module's parameters, are forgotten
    if not torch.cuda.is_current_stream_capturing():     # (nothing is released while a capture is under way)
        for k in [k for k, e in st["entries"].items() if e["used_step"] < st["step"] - 2 or e["storage"] not in st["storages"]]:
            del st["entries"][k]
    live = list(st["entries"].values())
    if not live:
        return
    key = tuple(id(e) for e in live)
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and __import__("os").environ.get("BEVMSDA_IMAGE_BATCH_CAPTURE", "1") == "0":
        return              # (A/B knob: a captured step packs image by image)
    if st["table_key"] != key:
        if capturing:
            return          # (no host-to-device copy inside a capture: this step packs image by image, as before round 6)
        lib = _lib.load()
        rows, first = [], 0
        for e in live:
            nb = lib.bevmsda_linear_pack_job_blocks(e["N"], e["K"], e["kind"])
            if nb <= 0:
                return
            # struct bevmsda_pack_job as five int64 words: w, ldw, blob, (N | K << 32), (kind | first_block << 32)
            rows.append([e["tensor"].data_ptr(), e["ldw"], e["blob"].data_ptr(), e["N"] | (e["K"] << 32), e["kind"] | (first << 32)])
            first += nb
        st["table"] = torch.tensor(rows, dtype=torch.int64).to(live[0]["blob"].device)
        st["table_key"], st["table_blocks"] = key, first
    with torch.cuda.device(st["table"].device):
        _lib.check(_lib.load().bevmsda_linear_pack_weights_multi_f32(
            st["table"].data_ptr(), len(live), st["table_blocks"], torch.cuda.current_stream().cuda_stream),
            "linear_pack_weights_multi")
    st["multi_launches"] += 1
    for e in live:
        e["fresh_step"] = st["step"]
        e["version"] = _ver(e["tensor"])


def end_training_steps():
    """A forward WITHOUT gradients started: weight images come from the per-version caches again."""
    _TRAIN["active"] = False


def clear_weight_caches(module):
    """Drop every derived weight image (packed / panel / transposed copies, cached on the parameters per version) of
    ``module``'s parameters.  Needed in ONE situation: parameters that are inference tensors (a model built or loaded
    under ``torch.inference_mode``) track no version, so an in-place write to them (``load_state_dict`` under
    ``inference_mode``) cannot invalidate the images — call this after such a write.  Returns the number dropped."""
    n = 0
    for p in module.parameters():
        for name in ("_bevmsda_pack", "_bevmsda_panel", "_bevmsda_wt"):
            if hasattr(p, name):
                delattr(p, name)
                n += 1
    for m in module.modules():          # merged-projection views cached on the owning modules (merged_linear_params)
        for name in [k for k in vars(m) if k.startswith("_merged_")]:
            delattr(m, name)
            n += 1
    return n


def _is_transposed_view(w):
    """(N, K) tensor whose memory is the row-major (K, N) matrix (``m.t()`` of a matrix with unit column stride)."""
    return w.dim() == 2 and w.shape[0] > 1 and w.shape[1] > 1 and w.stride(0) == 1 and w.stride(1) >= w.shape[0]


def packed_weight(weight):
    """Pre-split bf16 image of an (N, K) fp32 weight (``bevmsda_linear_pack_weight_f32``),
    cached on the tensor object until it is written to or moved.  A transposed view (``_is_transposed_view``) is packed
    from the memory it aliases (``bevmsda_linear_pack_weight_t_f32``)."""
    fresh = _train_image("pack", weight)
    if fresh is not None:
        return fresh
    key = (_ver(weight), weight.data_ptr(), tuple(weight.shape), weight.stride(0), weight.stride(1))
    hit = getattr(weight, "_bevmsda_pack", None)
    if hit is not None and hit[0] == key and _cache_ok(weight):
        _train_register("pack", weight, hit[1], launched=False)     # (a training step adopts it: rebuilt in the batch from now on)
        return _cached_image(hit, weight)
    lib = _lib.load()
    N, K = weight.shape
    nbytes = lib.bevmsda_linear_packed_bytes(N, K)
    if nbytes == 0:
        return None
    blob = torch.empty(nbytes // 2, dtype=torch.int16, device=weight.device)
    with torch.cuda.device(weight.device):
        if _is_transposed_view(weight):
            rc = lib.bevmsda_linear_pack_weight_t_f32(_ptr(weight), weight.stride(1), N, K, _ptr(blob),
                                                      torch.cuda.current_stream().cuda_stream)
        else:
            rc = lib.bevmsda_linear_pack_weight_f32(_ptr(weight), weight.stride(0), N, K, _ptr(blob),
                                                    torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "linear_pack_weight")
    try:
        weight._bevmsda_pack = (key, blob)
    except AttributeError:
        pass
    _train_register("pack", weight, blob)
    return blob


def panel_weight(weight):
    """Fragment-order bf16 image of an (N, K) fp32 weight for the row-panel kernel
    (``bevmsda_linear_panel_pack_weight_f32``), cached on the tensor until it is written to or moved."""
    fresh = _train_image("panel", weight)
    if fresh is not None:
        return fresh
    key = (_ver(weight), weight.data_ptr(), tuple(weight.shape), weight.stride(0), weight.stride(1))
    hit = getattr(weight, "_bevmsda_panel", None)
    if hit is not None and hit[0] == key and _cache_ok(weight):
        _train_register("panel", weight, hit[1], launched=False)
        return _cached_image(hit, weight)
    lib = _lib.load()
    N, K = weight.shape
    nbytes = lib.bevmsda_linear_panel_packed_bytes(N, K)
    if nbytes == 0:
        return None
    blob = torch.empty(nbytes // 2, dtype=torch.int16, device=weight.device)
    with torch.cuda.device(weight.device):
        if _is_transposed_view(weight):     # (the image of W^T from W where it lies)
            rc = lib.bevmsda_linear_panel_pack_weight_t_f32(_ptr(weight), weight.stride(1), N, K, _ptr(blob),
                                                            torch.cuda.current_stream().cuda_stream)
        else:
            rc = lib.bevmsda_linear_panel_pack_weight_f32(_ptr(weight), weight.stride(0), N, K, _ptr(blob),
                                                          torch.cuda.current_stream().cuda_stream)
    if rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_MISALIGNED):
        return None
    _lib.check(rc, "linear_panel_pack_weight")
    try:
        weight._bevmsda_panel = (key, blob)
    except AttributeError:
        pass
    _train_register("panel", weight, blob)
    return blob


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
    ``gather = (idx (M, 2) int32, scale (M,))``, SpatialCrossAttention's camera mean over the rows.  ``fc1`` / ``fc2``:
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
        if idx.dim() != 2 or idx.shape[1] != 2 or idx.dtype != torch.int32:
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


def transposed_weight(weight):
    """``weight.t()`` as the operand of an input-gradient GEMM: with weight packing on, the VIEW (``packed_weight`` builds
    the MFMA image straight from the aliased memory; ``linear`` copies it only for a kernel that wants the matrix);
    otherwise a contiguous copy cached on the tensor until it is written to."""
    if weight.is_cuda and weight.dim() == 2 and weight.stride(1) == 1 and _m().gemm_pack and _m().gemm != "native" \
            and _m().weight_views and weight.data_ptr() % 4 == 0:
        return weight.detach().t()
    key = (_ver(weight), weight.data_ptr(), tuple(weight.shape))
    hit = getattr(weight, "_bevmsda_wt", None)
    if hit is not None and hit[0] == key and _cache_ok(weight):
        return _cached_image(hit)
    with torch.no_grad():
        wt = weight.detach().t().contiguous()
    try:
        weight._bevmsda_wt = (key, wt)
    except AttributeError:
        pass
    return wt




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


def _adjacent(tensors):
    """Do the tensors lie back to back in one storage (``flatten_linear_params``), so that their row-wise concatenation is
    a view?"""
    t0 = tensors[0]
    if not all(t.is_contiguous() and t.dtype == t0.dtype and t.device == t0.device and t.shape[1:] == t0.shape[1:]
               for t in tensors):
        return False
    st = t0.untyped_storage().data_ptr()
    end = t0.data_ptr() + t0.numel() * t0.element_size()
    for t in tensors[1:]:
        if t.untyped_storage().data_ptr() != st or t.data_ptr() != end:
            return False
        end += t.numel() * t.element_size()
    return True


def flatten_linear_params(*linears):
    """Re-seat the weights (and biases) of ``nn.Linear`` layers that share their input back to back in ONE buffer each
    (``p.data`` becomes a view of it; values, Parameter objects, state_dict keys and optimizer state are untouched):
    their concatenation — the operand of the merged projection — is then a VIEW (``merged_linear_params``), where a
    training step paid a ``cat`` per group and step plus the copies of its backward.  Idempotent; a later ``.to()`` /
    ``.float()`` of the module gives every parameter its own storage again and the merge falls back to ``cat``."""
    ws, bs = [m.weight for m in linears], [m.bias for m in linears]
    if any(b is None for b in bs) or len({w.shape[1] for w in ws}) != 1:
        return False
    with torch.no_grad():
        for group in (ws, bs):
            if _adjacent([p.data for p in group]):
                continue
            flat = torch.cat([p.data for p in group], 0)
            o = 0
            for p in group:
                n = p.shape[0]
                p.data = flat[o:o + n]
                o += n
    return True


class _MergedParams(torch.autograd.Function):
    """(cat of the weights, cat of the biases) as VIEWS of the buffers the parameters were flattened into; the
    backward hands every parameter its block of the merged gradient, a view as well."""

    @staticmethod
    def forward(ctx, nw, *params):
        ws, bs = params[:nw], params[nw:]
        ctx.rows = [w.shape[0] for w in ws]
        K = ws[0].shape[1]
        N = sum(ctx.rows)
        w = ws[0].detach().as_strided((N, K), (K, 1))
        b = bs[0].detach().as_strided((N,), (1,))
        return w, b

    @staticmethod
    def backward(ctx, gw, gb):
        out_w, out_b, o = [], [], 0
        for n in ctx.rows:
            out_w.append(None if gw is None else gw[o:o + n])
            out_b.append(None if gb is None else gb[o:o + n])
            o += n
        return (None, *out_w, *out_b)


def merged_linear_params(owner, *linears, slot="_merged_linear"):
    """``cat`` of the weights / biases of ``nn.Linear`` layers that share their input (the
    sampling-offset and attention-weight projections; the value projections of all encoder
    layers), cached on ``owner`` while nothing needs a gradient and the parameters have not
    been written to.  Under autograd: a view when the parameters were laid out back to back
    (``flatten_linear_params``), else ``torch.cat``."""
    if torch.is_grad_enabled() and any(p.requires_grad for m in linears for p in (m.weight, m.bias)):
        ws, bs = [m.weight for m in linears], [m.bias for m in linears]
        if all(b is not None for b in bs) and _adjacent(ws) and _adjacent(bs):
            return _MergedParams.apply(len(ws), *ws, *bs)
        return (torch.cat(ws, 0), torch.cat(bs, 0))
    key = tuple((_ver(m.weight), _ver(m.bias), m.weight.data_ptr(), m.bias.data_ptr())
                for m in linears)
    hit = owner.__dict__.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    with torch.no_grad():
        w = torch.cat([m.weight for m in linears], 0)
        b = torch.cat([m.bias for m in linears], 0)
    owner.__dict__[slot] = (key, w, b)
    return w, b
