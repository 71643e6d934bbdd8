"""Weight images of the MFMA kernels (``packed_weight`` / ``panel_weight`` / ``transposed_weight``) and everything that keeps them
right: the per-tensor caches, the registry of images frozen into captured graphs, the one-launch rebuild of a training step's
images, merged and flattened projection parameters."""
import torch

from .. import _lib
from ..ext import _ptr
from ._base import _m, _ver


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
