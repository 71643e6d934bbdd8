"""BEVFormerEncoder / BEVFormerLayer on the MI355X kernels.

Drop-in for the reference classes of the same registry names
(projects/mmdet3d_plugin/bevformer/modules/encoder.py:24-239 and :242-406):
same constructor arguments (built from the ``encoder=dict(...)`` block of the
configs, bevformer_base.py:78-105), same forward keyword contract (call site
modules/transformer.py:186-198), same ``state_dict`` keys.

What is organised differently (results unchanged):
  * all per-frame geometry — reference points, camera projection, visibility,
    the ragged (camera, query) row list of SCA, the camera-count reciprocal and
    the small (1,2)/(1,) BEV shape tensors the reference re-creates in every
    layer (encoder.py:370-372) — is built once per call into a ``FramePlan``
    and cached across calls while the camera matrices do not change;
  * optional BEV-query tiling over the GPUs of a node (``bev_tiling.py``): each
    rank runs the layer stack on a contiguous block of BEV rows and the grid is
    reassembled with an RCCL all-gather.
"""
import copy
import warnings

import torch

from ..registry import (TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, BaseModule, ModuleList,
                        auto_fp16, build_transformer_layer, force_fp32)
from . import geometry
from .. import ops
from .bricks import FFN
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer


class TransformerLayerSequence(BaseModule):
    """mmcv's ``TransformerLayerSequence`` surface: ``num_layers`` deep copies
    of the layer config -> ``layers``; ``embed_dims`` / ``pre_norm`` mirrored
    from the first layer (read by modules/transformer.py)."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


@TRANSFORMER_LAYER_SEQUENCE.register_module(force=True)
class BEVFormerEncoder(TransformerLayerSequence):

    def __init__(self, *args, pc_range=None, num_points_in_pillar=4, return_intermediate=False,
                 dataset_type="nuscenes", **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.num_points_in_pillar = num_points_in_pillar
        self.pc_range = pc_range
        self.fp16_enabled = False
        self._plan_cache = {}
        self.plan_cache_size = 4
        # order of the ragged SCA rows inside a camera: "image" (Z-order of the projected pillar: rows
        # handled together sample neighbouring pixels — best for the forward kernels (235 vs 289 us) and,
        # since the backward sorts its taps per workgroup in LDS, for the backward too (1.39 vs 1.56 ms;
        # the first-generation atomic backward preferred raster rows: r1t profiles), "raster" (BEV
        # order, the reference's nonzero() order) or "polar"; "auto" = image
        self.sca_row_order = "auto"
        self.bev_tiling = None          # set by bev_tiling.enable_bev_tiling()
        # GPU tensors: plans come from the HIP frame-plan kernels (geometry.DevicePlanner: no host
        # sync, row count on the device); False keeps the torch-op builder with its host syncs
        self.device_plans = True
        self._planners = {}
        # inference (``ops.using(overlap_value_proj=...)`` / BEVMSDA_OVERLAP at import; on by default since round 6): the hoisted
        # SCA value projection is issued on a second stream, so that it runs beside the TSA value projection and the first
        # layer's TemporalSelfAttention chain, joined before the first SpatialCrossAttention.  Two chip-filling kernels do not
        # interleave (§4 "co-scheduling"), but each of these launches ends in a partial round of workgroups (the camera-value
        # GEMM: 5.6 rounds of one 128-row workgroup per CU) and the other stream's workgroups fill those tails: 4.00-4.04 against
        # 4.09-4.11 ms per base frame, four interleaved rounds (profiles/r6/r6z_overlap_ab.txt; round 2, on slower kernels: 0.8 %).
        self.overlap_value_proj = None      # None: the ``overlap_value_proj`` mode (bevformer_amd/modes.py)
        self._side_stream = None

    # kept as static/instance methods with the reference's names and outputs
    get_reference_points = staticmethod(geometry.get_reference_points)

    @force_fp32(apply_to=("reference_points", "img_metas"))
    def point_sampling(self, reference_points, pc_range, img_metas):
        return geometry.point_sampling(reference_points, pc_range, img_metas)

    def row_order(self):
        if self.sca_row_order != "auto":
            return self.sca_row_order
        return "image"

    def frame_plan(self, bev_h, bev_w, bs, img_metas, device, dtype, tile=None, cell_perm=None, train_fast=None):
        """The per-frame geometry.  On a GPU: two kernel launches into the planner's buffers, no
        host synchronisation (under autograd the plan is then materialised: one read of the row
        count, the torch statements of that path need sizes).  ``tile = (q0, q1)``: rows only for
        those BEV queries (bev_tiling).  ``cell_perm = (name, perm)``: the plan's queries are the BEV cells in that order
        (bev_tiling's sector layout; ``name`` keys the caches).  ``train_fast``: the caller's decision whether this call
        runs the autograd fast path (``forward`` decides it once from its tensors; None: decided here from ``dtype``)."""
        device = torch.device(device)
        order = self.row_order()
        pname, perm = cell_perm if cell_perm is not None else (None, None)
        if self.device_plans and device.type == "cuda":
            num_cams = len(img_metas[0]["lidar2img"])
            key = (bev_h, bev_w, bs, str(device), order, tile, num_cams, self.num_points_in_pillar, pname)
            planner = self._planners.get(key)
            if planner is None:
                if len(self._planners) >= 8:
                    self._planners.pop(next(iter(self._planners)))
                planner = self._planners[key] = geometry.DevicePlanner(
                    bev_h, bev_w, bs, self.pc_range, self.num_points_in_pillar, num_cams, device,
                    row_order=order, tile=tile, cell_perm=perm)
            plan = planner.plan(img_metas)
            if not torch.is_grad_enabled():
                return plan
            # autograd: the fast path (train_ops.py) keeps the row count on the device and works on a private copy of
            # the row arrays; the per-op path needs sizes on the host (one read of the counters)
            fast = self._train_fast_path(device) and dtype == torch.float32 if train_fast is None else train_fast
            return plan.snapshot() if fast else plan.materialize()
        assert tile is None, "tiles of a host-built plan come from bev_tiling.slice_plan"
        assert order in geometry.ROW_ORDERS, f"host-built plans know the row orders {geometry.ROW_ORDERS}"
        key = geometry.plan_key(bev_h, bev_w, bs, self.pc_range, self.num_points_in_pillar,
                                img_metas, device, dtype) + (order, pname)
        plan = self._plan_cache.get(key)
        if plan is None:
            plan = geometry.build_frame_plan(bev_h, bev_w, bs, self.pc_range,
                                             self.num_points_in_pillar, img_metas, device, dtype,
                                             row_order=order, cell_perm=perm)
            if len(self._plan_cache) >= self.plan_cache_size:
                self._plan_cache.pop(next(iter(self._plan_cache)))
            self._plan_cache[key] = plan
        return plan

    def _train_fast_path(self, device=None, tensors=()):
        """Gradients are recorded and every layer can run its row-local parts on the chain kernels (train_ops.py):
        the reference's operation order, the MFMA kernels in use, every parameter (and every tensor of ``tensors``:
        the call's floating-point inputs) fp32 — a ``.half()`` / ``.bfloat16()`` / ``.double()`` model or input takes
        the per-op path, whose statements fall back to torch for the dtypes the kernels do not cover."""
        if not (torch.is_grad_enabled() and ops.modes().train_chain and not torch.is_autocast_enabled()
                and ops.gemm_mode() != "native" and (device is None or torch.device(device).type == "cuda")):
            return False
        if any(t is not None and torch.is_tensor(t) and t.is_floating_point() and t.dtype != torch.float32
               for t in tensors):
            return False
        if any(p.dtype != torch.float32 for p in self.parameters()):
            return False
        return all(getattr(layer, "chain_trainable", lambda: False)() for layer in self.layers)

    def _flatten_projection_params(self):
        """Training fast path, once: lay the parameters of every merged projection back to back in memory
        (``ops.flatten_linear_params``) — the value projections of all layers (one grouped GEMM per family) and each
        attention's sampling-offset + attention-weight pair — so that their concatenations are views."""
        if not ops.modes().flatten_params or torch.cuda.is_current_stream_capturing():
            return
        atts = [getattr(layer, "attentions", None) for layer in self.layers]
        if any(a is None or len(a) != 2 or not hasattr(a[1], "deformable_attention") for a in atts):
            return
        tsas = [a[0] for a in atts]
        scas = [a[1].deformable_attention for a in atts]
        if self.__dict__.get("_flat_params") and len(scas) > 1 \
                and ops._adjacent([m.value_proj.weight.data for m in scas]):
            return      # (still laid out back to back; a ``.to()`` / ``.float()`` / ``load_state_dict(assign=True)`` since
            #             the last call gave every parameter its own storage again: lay them out anew)
        ops.flatten_linear_params(*[m.value_proj for m in scas])
        ops.flatten_linear_params(*[m.value_proj for m in tsas])
        for m in tsas + scas:
            ops.flatten_linear_params(m.sampling_offsets, m.attention_weights)
        self.__dict__["_flat_params"] = True

    def _contiguous_pos(self, pos):
        """``pos`` (bs, Q, C) with unit stride along C.  The positional encoding of the BEV grid does not change from
        frame to frame: at inference the copy is kept for as long as the caller hands in the same (unmodified) memory —
        the cache holds the source's storage, so its address cannot be recycled for other values behind its back."""
        if pos.is_contiguous():
            return pos
        if (torch.is_grad_enabled() and pos.requires_grad) or pos.is_inference():
            return pos.contiguous()         # (inference tensors track no version counter: nothing to key a cache on)
        storage = pos.untyped_storage()
        key = (storage.data_ptr(), pos.storage_offset(), pos._version, tuple(pos.shape), tuple(pos.stride()), pos.dtype)
        hit = self.__dict__.get("_pos_cache")
        if hit is None or hit[0] != key:
            hit = (key, pos.detach().contiguous(), storage)
            self.__dict__["_pos_cache"] = hit
        return hit[1]

    def hoisted_value_projections(self, value, tsa_value, plan=None, spatial_shapes=None):
        """The layer-invariant projections, issued once for all layers.

        The inputs of ``MSDeformableAttention3D.value_proj`` (camera features,
        spatial_cross_attention.py:334) and — when a history BEV exists — of
        ``TemporalSelfAttention.value_proj`` (``[prev_bev, bev_query]`` built before the layer
        loop, encoder.py:204-209; temporal_self_attention.py:198) do not change from layer to
        layer; only the weights do.  The reference streams the 189 MB of camera features six
        times; here one grouped GEMM reads each input once and writes every layer's
        projected value (``ops.linear(groups=num_layers)``).  Inference path only.
        ``plan``: a TILE's device-side frame plan (BEV tiling over GPUs): cameras none of whose pixels the tile's
        queries can sample (no ragged row: a device-side count) are skipped by the camera-value projection.
        Returns (per-layer SCA values or None, per-layer TSA values or None)."""
        from .spatial_cross_attention import MSDeformableAttention3D, SpatialCrossAttention
        from .temporal_self_attention import TemporalSelfAttention
        if torch.is_grad_enabled() or self.training or ops.gemm_mode() == "native" \
                or not value.is_cuda or len(self.layers) < 2:
            return None, None
        tsas, scas = [], []
        for layer in self.layers:
            att = getattr(layer, "attentions", None)
            if att is None or len(att) != 2 or not isinstance(att[0], TemporalSelfAttention) \
                    or not isinstance(att[1], SpatialCrossAttention) \
                    or not isinstance(att[1].deformable_attention, MSDeformableAttention3D):
                return None, None
            tsas.append(att[0])
            scas.append(att[1].deformable_attention)
        L = len(self.layers)
        sca_vals = tsa_vals = None
        Nc, S, bs, C = value.shape
        feats = value.permute(2, 0, 1, 3).reshape(bs * Nc, S, C)
        w, b = ops.merged_linear_params(self, *[m.value_proj for m in scas], slot="_merged_sca_value")
        store = ops.value_storage()          # bf16 storage: the GEMM rounds its fp32 result on the way out
        self._sca_ready = None
        seg = None
        if plan is not None and getattr(plan, "dynamic", False) and plan.cam_start is not None \
                and plan.cam_start.numel() == bs * Nc + 1 \
                and spatial_shapes is not None and spatial_shapes.dtype == torch.long and spatial_shapes.is_cuda:
            # rows of (batch entry, camera) = one segment of S feature rows (+ the sampling kernel's zero-weight taps,
            # which reach one image row + 1 pixel into the neighbouring cameras' rows: ops.linear docstring)
            seg = (plan.cam_start, S, spatial_shapes.contiguous())
        self._last_segments = seg            # (bench.py reports how many cameras a rank projects)
        overlap = ops.modes().overlap_value_proj if self.overlap_value_proj is None else self.overlap_value_proj
        if overlap and not ops.gemm_timer_active():     # (a recording timer brackets launches with events on ONE stream)
            cur = torch.cuda.current_stream(value.device)
            if self._side_stream is None or self._side_stream.device != value.device:
                self._side_stream = torch.cuda.Stream(value.device)
            side = self._side_stream
            ops.packed_weight(w)                       # (weight images built on the main stream, once)
            if w.is_contiguous():
                ops.panel_weight(w)
            side.wait_stream(cur)
            feats.record_stream(side)                  # (bs > 1: a copy made on the main stream, read by the side stream's kernel)
            with torch.cuda.stream(side):
                y = ops.linear(feats, w, b, groups=L, out_dtype=store, tag="sca_value_proj", segments=seg)
                if y is not None:
                    self._sca_ready = side.record_event()
                    y.record_stream(cur)
        else:
            y = ops.linear(feats, w, b, groups=L, out_dtype=store, tag="sca_value_proj", segments=seg)
        if y is not None:
            M = scas[0].num_heads
            sca_vals = [y[i].view(bs * Nc, S, M, -1) for i in range(L)]
            if seg is not None:
                # segments of cameras this rank cannot see were left UNWRITTEN (torch.empty): only the fused D = 32
                # sampling kernel, which touches at most max W + 1 rows past a used camera, may read these tensors —
                # any other consumer (SpatialCrossAttention's fallback paths) projects the features itself
                for v in sca_vals:
                    v._bevmsda_partial = True
        if tsa_value is not None:
            w, b = ops.merged_linear_params(self, *[m.value_proj for m in tsas], slot="_merged_tsa_value")
            if isinstance(tsa_value, tuple):
                # (history (1, Q, C), current (1, Q, C)): the two row blocks of stack([prev_bev, bev_query]) read where
                # they lie (ops.linear_rows2) — the stack itself is 82 MB written and read per base frame
                hist, cur = tsa_value
                y = ops.linear_rows2(hist.reshape(-1, hist.shape[-1]), cur.reshape(-1, cur.shape[-1]), w, b, groups=L,
                                     out_dtype=store, tag="tsa_value_proj")
                nb, nv = 2, hist.shape[1]
                if y is None:
                    y = ops.linear(torch.stack([hist, cur], 1).reshape(2, nv, -1), w, b, groups=L, out_dtype=store,
                                   tag="tsa_value_proj")
            else:
                y = ops.linear(tsa_value, w, b, groups=L, out_dtype=store, tag="tsa_value_proj")
                nb, nv = tsa_value.shape[0], tsa_value.shape[1]
            if y is not None:
                M = tsas[0].num_heads
                tsa_vals = [y[i].view(nb, nv, M, -1) for i in range(L)]
        return sca_vals, tsa_vals

    def tsa_seam(self, li, first, pos):
        """The offer layer ``li`` gets to run the NEXT layer's TemporalSelfAttention offset / weight projection behind its own
        last kernel (``ops.proj_ffn_chain(tail=...)``, csrc/linear_chain.h TP): that projection reads ``cat([first, query +
        pos], -1)`` (temporal_self_attention.py:197-211) with query = the rows layer ``li`` is about to produce, so the
        workgroup that holds them projects them on the spot — one launch and one read of the grid less per layer.  ``first``
        (1, Q, 256): the history rows of the queries, ``pos`` (1, Q, 256).  Inference at bs = 1 between two stock layers only;
        returns the holder ``{"first", "pos", "w", "b", "proj": None}`` (the layer fills ``proj``) or None."""
        from .temporal_self_attention import TemporalSelfAttention
        if li + 1 >= len(self.layers) or not ops.modes().tsa_seam or torch.is_grad_enabled() or self.training \
                or first is None or pos is None or not first.is_cuda or first.dim() != 3 or first.shape[0] != 1 \
                or first.shape != pos.shape or first.shape[-1] != 256 or first.dtype != torch.float32 \
                or pos.dtype != torch.float32 or ops.gemm_mode() == "native":
            return None
        cur, nxt = self.layers[li], self.layers[li + 1]
        if type(cur) is not BEVFormerLayer or type(nxt) is not BEVFormerLayer or cur.pre_norm or nxt.pre_norm \
                or tuple(cur.operation_order[-4:]) != ("cross_attn", "norm", "ffn", "norm") \
                or nxt.operation_order[0] != "self_attn":
            return None
        tsa = nxt.attentions[0]
        if type(tsa) is not TemporalSelfAttention or not tsa.batch_first or tsa.num_bev_queue != 2 or tsa.embed_dims != 256:
            return None
        w, b = ops.merged_linear_params(tsa, tsa.sampling_offsets, tsa.attention_weights)
        if tuple(w.shape) != (w.shape[0], 512) or w.shape[0] % 64 or w.shape[0] > 256:
            return None
        return {"first": first, "pos": pos, "w": w, "b": b, "proj": None}

    def _stack_free(self, history, bev_query, bs):
        """May the layers run without the stacked [history, bev_query] tensor?  Inference at bs = 1 on the GPU with every
        layer a stock ``BEVFormerLayer`` whose first attention is the stock ``TemporalSelfAttention`` (a subclass or another
        module may read ``value`` / ``prev_bev``)."""
        from .temporal_self_attention import TemporalSelfAttention
        if bs != 1 or not ops.modes().stack_free or torch.is_grad_enabled() or self.training or not history.is_cuda \
                or len(self.layers) < 2 \
                or ops.gemm_mode() == "native" or history.dtype != torch.float32 or bev_query.dtype != torch.float32 \
                or history.shape != bev_query.shape:
            return False
        return all(type(layer) is BEVFormerLayer and type(getattr(layer, "attentions", [None])[0]) is TemporalSelfAttention
                   for layer in self.layers)

    def hoisted_value_projections_autograd(self, value, tsa_value):
        """``hoisted_value_projections`` with gradients (train_ops.grouped_linear): camera features (Nc, S, bs, C) ->
        per-layer (bs * Nc, S, M, D) values; ``tsa_value``: None, the stacked (bs * 2, Q, C) tensor, or (bs = 1) the
        pair (history (1, Q, C), current (1, Q, C)) -> per-layer (bs * 2, Q, M, D) values."""
        from .. import train_ops
        scas = [layer.attentions[1].deformable_attention for layer in self.layers]
        tsas = [layer.attentions[0] for layer in self.layers]
        L = len(self.layers)
        Nc, S, bs, C = value.shape
        if C != 256 or any(m.value_proj.weight.shape != (256, 256) for m in scas + tsas):
            return None, None
        srcs = [value] + ([] if tsa_value is None else list(tsa_value) if isinstance(tsa_value, tuple) else [tsa_value])
        srcs += [p for m in scas + tsas for p in (m.value_proj.weight, m.value_proj.bias)]
        if any(t is None or not t.is_cuda or t.dtype != torch.float32 for t in srcs):
            return None, None           # (the grouped GEMM is an fp32 kernel: other dtypes take the per-op path)
        feats = value.permute(2, 0, 1, 3).reshape(bs * Nc, S, C)
        w, b = ops.merged_linear_params(self, *[m.value_proj for m in scas], slot="_merged_sca_value")
        # bf16 value storage: the GEMM rounds in its epilogue, the sampling Functions hand their fp32 gradients back
        # through a sink (no conversion pass in either direction)
        store = ops.value_storage()
        bf = store == torch.bfloat16
        # the consumers' value gradients land side by side in one array per family (train_ops.ValueGradSink)
        self._value_sinks = (train_ops.ValueGradSink(L), train_ops.ValueGradSink(L))
        ys = train_ops.grouped_linear(feats, w, b, L, "sca_value_proj", out_dtype=store if bf else None,
                                      sink=self._value_sinks[0])
        M = scas[0].num_heads
        sca_vals = [y.view(bs * Nc, S, M, -1) for y in ys]
        tsa_vals = None
        if tsa_value is not None:
            w, b = ops.merged_linear_params(self, *[m.value_proj for m in tsas], slot="_merged_tsa_value")
            if isinstance(tsa_value, tuple):
                Q = tsa_value[0].shape[1]
                ys = train_ops.grouped_linear([t.reshape(-1, C) for t in tsa_value], w, b, L, "tsa_value_proj",
                                              out_dtype=store if bf else None, sink=self._value_sinks[1])
                nb = 2
            else:
                Q = tsa_value.shape[1]
                nb = tsa_value.shape[0]
                ys = train_ops.grouped_linear(tsa_value, w, b, L, "tsa_value_proj", out_dtype=store if bf else None,
                                              sink=self._value_sinks[1])
            M = tsas[0].num_heads
            tsa_vals = [y.view(nb, Q, M, -1) for y in ys]
        return sca_vals, tsa_vals

    @auto_fp16()
    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, prev_bev=None,
                shift=0.0, **kwargs):
        """bev_query / bev_pos / prev_bev (Q, bs, C); key = value (Nc, S, bs, C);
        returns (bs, Q, C), or (num_layers, bs, Q, C) with return_intermediate."""
        lowp = [t for t in (bev_query, key, value, bev_pos, prev_bev)
                if torch.is_tensor(t) and t.dtype in (torch.float16, torch.bfloat16)]
        if lowp:
            # fp16 wrapper of the reference (``@auto_fp16()`` at encoder.py:151 with ``fp16_enabled`` set by
            # ``wrap_fp16_model``, tools/fp16/train.py:224-226; ``get_bev_features`` hands over half ``bev_queries`` /
            # ``prev_bev`` / ``bev_pos``, transformer.py:103): the inputs arrive ROUNDED to half precision; the kernels of
            # this package compute in fp32 (bf16 storage at most), so the rounded inputs are widened once and the
            # layer stack runs its fp32 path with autocast off — at least the reference's arithmetic (its Linear layers
            # run in fp16 under autocast, its sampling and SpatialCrossAttention in fp32: ``custom_fwd(cast_inputs=
            # torch.float32)``, ``force_fp32`` at spatial_cross_attention.py:75) — and returns fp32 like the
            # reference's last LayerNorm does under autocast
            def up(t):
                return t.float() if torch.is_tensor(t) and t.dtype in (torch.float16, torch.bfloat16) else t
            dev_type = bev_query.device.type
            with torch.autocast(dev_type, enabled=False):
                same = key is value
                value = up(value)
                return self._forward(up(bev_query), value if same else up(key), value, *[up(a) for a in args],
                                     bev_h=bev_h, bev_w=bev_w, bev_pos=up(bev_pos), spatial_shapes=spatial_shapes,
                                     level_start_index=level_start_index, valid_ratios=valid_ratios, prev_bev=up(prev_bev),
                                     shift=up(shift), **kwargs)
        return self._forward(bev_query, key, value, *args, bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos,
                             spatial_shapes=spatial_shapes, level_start_index=level_start_index, valid_ratios=valid_ratios,
                             prev_bev=prev_bev, shift=shift, **kwargs)

    def _forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                 spatial_shapes=None, level_start_index=None, valid_ratios=None, prev_bev=None,
                 shift=0.0, **kwargs):
        """``forward`` behind the precision wrapper (fp32 tensors, or whatever dtype an fp32-unaware caller chose)."""
        if self.bev_tiling is not None:
            from .. import bev_tiling
            return bev_tiling.tiled_forward(self, bev_query, key, value, *args, bev_h=bev_h,
                                            bev_w=bev_w, bev_pos=bev_pos,
                                            spatial_shapes=spatial_shapes,
                                            level_start_index=level_start_index,
                                            prev_bev=prev_bev, shift=shift, **kwargs)
        if torch.is_grad_enabled() and bev_query.is_cuda and value.is_cuda:
            # a differentiable step: ONE launch rebuilds the weight images the last steps used from the weights' current
            # values (ops.begin_training_step; until round 6 every image was re-packed by its own launch, 52 per step at base)
            ops.begin_training_step(self)
        else:
            ops.end_training_steps()
        bs = bev_query.size(1)
        train_fast = bev_query.is_cuda and value.is_cuda and self._train_fast_path(
            value.device, (bev_query, value, bev_pos, prev_bev))
        overlap = ops.modes().overlap_value_proj if self.overlap_value_proj is None else self.overlap_value_proj
        plan_ready = None
        if overlap and ops.modes().plan_on_side and not torch.is_grad_enabled() and bev_query.is_cuda and value.is_cuda \
                and self.device_plans and not ops.gemm_timer_active():
            # the plan kernels go first on the stream that will carry the hoisted camera-value projection: the main stream
            # starts the BEV-value projection at once and meets the plan where it meets the projected camera values — at
            # the first SpatialCrossAttention (``_sca_ready``, recorded behind both); the side stream waits for the main
            # one first, i.e. for the previous frame's readers of the planner's buffers
            cur = torch.cuda.current_stream(value.device)
            if self._side_stream is None or self._side_stream.device != value.device:
                self._side_stream = torch.cuda.Stream(value.device)
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                plan = self.frame_plan(bev_h, bev_w, bs, kwargs["img_metas"], bev_query.device,
                                       bev_query.dtype, train_fast=train_fast)
                plan_ready = self._side_stream.record_event()
        else:
            plan = self.frame_plan(bev_h, bev_w, bs, kwargs["img_metas"], bev_query.device,
                                   bev_query.dtype, train_fast=train_fast)
        ref_2d = plan.ref_2d

        bev_query = bev_query.permute(1, 0, 2)
        # (through get_bev_features the positional encoding arrives as a transposed view of (bs, C, H*W): made contiguous
        # HERE, once per frame — left to the kernels' argument checks it was copied by every layer, 60 us each)
        bev_pos = self._contiguous_pos(bev_pos.permute(1, 0, 2))
        len_bev = ref_2d.shape[1]
        history = None
        if prev_bev is not None:
            prev_bev = prev_bev.permute(1, 0, 2)
            history = prev_bev
            prev_bev = None                 # stack([history, bev_query]): built below, only if something reads it
            hybird_ref_2d = geometry.hybrid_ref_2d(ref_2d, shift)      # (= stack([ref_2d + shift, ref_2d], 1), one launch)
        else:
            hybird_ref_2d = geometry.hybrid_ref_2d(ref_2d, None)       # (= stack([ref_2d, ref_2d], 1))

        # TSA's value is stack([history, bev_query]).  Inference, bs = 1: the grouped value projection reads the two tensors
        # where they lie and every layer's TSA gets its projected value and the history rows — nothing reads the stacked
        # tensor (82 MB written + read per base frame), a (2, Q, C) VIEW of the history stands in for it (value[:1] IS the
        # history; TemporalSelfAttention reads nothing else of it when ``tsa_projected_value`` is given)
        stack_free = history is not None and self._stack_free(history, bev_query, bs)
        if history is not None and not stack_free:
            prev_bev = torch.stack([history, bev_query], 1).reshape(bs * 2, len_bev, -1)
        sca_vals, tsa_vals = self.hoisted_value_projections(value, (history, bev_query) if stack_free else prev_bev)
        if plan_ready is not None and getattr(self, "_sca_ready", None) is None:
            torch.cuda.current_stream(value.device).wait_event(plan_ready)      # (no side-stream projection to meet: join now)
        if stack_free:
            prev_bev = history.expand(2, len_bev, history.shape[-1]) if tsa_vals is not None \
                else torch.stack([history, bev_query], 1).reshape(bs * 2, len_bev, -1)
        import contextlib
        with contextlib.ExitStack() as scope:
            share = None
            fast_train = False
            if sca_vals is None and tsa_vals is None and train_fast:
                self._flatten_projection_params()
                from .. import train_ops
                # the parameter-gradient arena of THIS call's backward pass: it travels in the modes snapshot every
                # autograd Function of the step takes in its forward (no module-global state: two encoders, or two
                # threads, never share accumulators)
                scope.enter_context(ops.using(grad_arena=train_ops.begin_step(
                    sum(p.numel() for p in self.parameters() if p.requires_grad))))
                # autograd fast path: the same two grouped GEMMs as autograd Functions (their backward sums the six input
                # gradients in the GEMM epilogues); with bs = 1 the history BEV and the current queries stay two tensors
                # (no gradient is formed for a detached history)
                sca_vals, tsa_vals = self.hoisted_value_projections_autograd(
                    value, None if history is None else (history, bev_query) if bs == 1 else prev_bev)
                fast_train = sca_vals is not None
            if not fast_train and torch.is_grad_enabled() and len(self.layers) > 1 and value.is_cuda \
                    and ops.modes().grad_thread:
                # training: the camera features and [prev_bev, bev_query] feed every layer's value projection — their six
                # input gradients are summed inside the GEMMs instead of by autograd's adds (ops.GradThread)
                share = {"sca": ops.GradThread(), "tsa": ops.GradThread()}
            return self._run_layers(bev_query, key, value, args, kwargs, plan, bev_pos, hybird_ref_2d, bev_h, bev_w,
                                    spatial_shapes, level_start_index, prev_bev, history, share, fast_train, sca_vals,
                                    tsa_vals)

    def _run_layers(self, bev_query, key, value, args, kwargs, plan, bev_pos, hybird_ref_2d, bev_h, bev_w, spatial_shapes,
                    level_start_index, prev_bev, history, share, fast_train, sca_vals, tsa_vals):
        """The layer loop of ``forward`` (encoder.py:211-233)."""
        output = bev_query          # (zero layers: the queries come back unchanged, encoder.py:211)
        intermediate = []
        seam = None
        for li, layer in enumerate(self.layers):
            hoisted = {}
            if share is not None:
                hoisted["value_grad_share"] = share
            if (share is not None or fast_train) and history is not None:
                hoisted["tsa_history"] = history
            if fast_train and getattr(self, "_value_sinks", None) is not None:
                hoisted["projected_value_sink"] = (self._value_sinks[0], li)
                hoisted["tsa_projected_value_sink"] = (self._value_sinks[1], li)
            if sca_vals is not None:
                hoisted["projected_value"] = sca_vals[li]
                if li == 0 and getattr(self, "_sca_ready", None) is not None:
                    hoisted["projected_value_ready"] = self._sca_ready
            if tsa_vals is not None:
                hoisted["tsa_projected_value"] = tsa_vals[li]
            if seam is not None and seam.get("proj") is not None:
                hoisted["tsa_proj"] = seam["proj"]          # (made by the previous layer's last kernel)
            seam = self.tsa_seam(li, history, bev_pos) if history is not None else None
            if seam is not None:
                hoisted["tsa_seam"] = seam
            output = layer(bev_query, key, value, *args, bev_pos=bev_pos, ref_2d=hybird_ref_2d,
                           ref_3d=plan.ref_3d, bev_h=bev_h, bev_w=bev_w,
                           spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                           reference_points_cam=plan.reference_points_cam,
                           bev_mask=plan.bev_mask, prev_bev=prev_bev, frame_plan=plan, **hoisted,
                           **kwargs)
            bev_query = output
            if self.return_intermediate:
                intermediate.append(output)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output


@TRANSFORMER_LAYER.register_module(force=True)
class BEVFormerLayer(MyCustomBaseTransformerLayer):
    """One encoder layer: TSA -> LN -> SCA -> LN -> FFN -> LN in the order given
    by ``operation_order`` (encoder.py:243-406)."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"),
                 ffn_num_fcs=2, **kwargs):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")     # the reference config style is the deprecated one
            super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                             ffn_dropout=ffn_dropout, operation_order=operation_order,
                             act_cfg=act_cfg, norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs,
                             **kwargs)
        self.fp16_enabled = False
        assert len(operation_order) == 6
        assert set(operation_order) == set(["self_attn", "norm", "cross_attn", "ffn"])

    def chain_trainable(self):
        """The autograd path of this layer can run on the row-chain kernels (train_ops.py): the reference's operation
        order (post-norm), TemporalSelfAttention / SpatialCrossAttention over 256 channels, the 256 -> 512 -> 256 FFN,
        LayerNorms.  Active dropout (train() mode, p > 0 in the attentions / the FFN) is covered: the seam Functions draw
        the scale tensors of the reference's nn.Dropout sites and the chain kernels apply them."""
        from .spatial_cross_attention import MSDeformableAttention3D, SpatialCrossAttention
        from .temporal_self_attention import TemporalSelfAttention
        if tuple(self.operation_order) != ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm") or self.pre_norm:
            return False
        att = getattr(self, "attentions", None)
        if att is None or len(att) != 2 or len(self.ffns) != 1 or len(self.norms) != 3:
            return False
        t, s_, ffn = att[0], att[1], self.ffns[0]
        if not (isinstance(t, TemporalSelfAttention) and isinstance(s_, SpatialCrossAttention)
                and isinstance(s_.deformable_attention, MSDeformableAttention3D) and isinstance(ffn, FFN)):
            return False
        if t.embed_dims != 256 or s_.embed_dims != 256 or not t.batch_first or ffn.num_fcs != 2 or not ffn.add_identity \
                or ffn.feedforward_channels != 512 or not all(isinstance(n, torch.nn.LayerNorm) for n in self.norms):
            return False
        # the FFN as mmcv builds it: [Linear, ReLU, Dropout], Linear, Dropout (+ an identity dropout_layer)
        lay = list(ffn.layers)
        if len(lay) != 3 or not isinstance(lay[0][2], torch.nn.Dropout) or not isinstance(lay[2], torch.nn.Dropout) \
                or getattr(ffn.dropout_layer, "p", 0.0) > 0:
            return False
        return True

    def forward(self, query, key=None, value=None, bev_pos=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, ref_2d=None,
                ref_3d=None, bev_h=None, bev_w=None, reference_points_cam=None, mask=None,
                spatial_shapes=None, level_start_index=None, prev_bev=None, frame_plan=None,
                tsa_seam=None, tsa_proj=None, **kwargs):
        """``tsa_seam``: the encoder's offer to project the rows this layer produces for the next layer's
        TemporalSelfAttention in the layer's last kernel (``BEVFormerEncoder.tsa_seam``); ``tsa_proj``: that projection of
        THIS layer's input rows, made by the previous layer."""
        norm_i = attn_i = ffn_i = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
            warnings.warn(f"Use same attn_mask in all attentions in {self.__class__.__name__} ")
        else:
            assert len(attn_masks) == self.num_attn
        if frame_plan is not None:
            bev_shapes, bev_start = frame_plan.bev_shapes, frame_plan.bev_start
        else:
            bev_shapes = torch.tensor([[bev_h, bev_w]], device=query.device)
            bev_start = torch.tensor([0], device=query.device)

        # inference fast path: "+ identity" of an attention / FFN step is folded into the
        # LayerNorm that follows it (one pass over the grid instead of two launches)
        order = self.operation_order
        # (post-norm orders only: with a leading norm the "+ identity" of a step feeds the NEXT
        # step's residual, not the norm that follows)
        fuse_norm = (not self.training) and (not torch.is_grad_enabled()) and not self.pre_norm
        # autograd path: the same "+ identity" deferral, resolved by ops.add_layernorm_autograd (one forward pass,
        # one backward kernel instead of torch's add + LayerNorm and their four backward launches)
        fuse_norm_grad = torch.is_grad_enabled() and query.is_cuda and not self.pre_norm
        # ... and the row-chain kernels as the forward of autograd Functions that save what their backward needs
        fuse_chain_grad = fuse_norm_grad and ops.modes().train_chain and not torch.is_autocast_enabled() \
            and ops.gemm_mode() != "native" and self.chain_trainable()
        pending = None                          # (branch output, identity) awaiting its norm

        def _defer(i):
            return (fuse_norm or fuse_norm_grad) and i + 1 < len(order) and order[i + 1] == "norm" \
                and isinstance(self.norms[norm_i], torch.nn.LayerNorm)

        def _post_norm(i):
            """The LayerNorm the step at ``i`` may fold into its last projection (inference fast path)."""
            return self.norms[norm_i] if ((fuse_norm or fuse_chain_grad) and _defer(i)) else None

        def _chain(i):
            """cross_attn at ``i`` followed by norm, ffn, norm (the reference's order): a callable that runs the rest
            of the layer — output projection, "+ identity", norm, FFN, "+ identity", norm — in one kernel."""
            if not (fuse_norm or fuse_chain_grad) or tuple(order[i + 1:i + 4]) != ("norm", "ffn", "norm") \
                    or ffn_i >= len(self.ffns):
                return None
            ffn, n0, n1 = self.ffns[ffn_i], self.norms[norm_i], self.norms[norm_i + 1]
            if not isinstance(ffn, FFN) or ffn.num_fcs != 2 or not ffn.add_identity \
                    or not isinstance(n0, torch.nn.LayerNorm) or not isinstance(n1, torch.nn.LayerNorm):
                return None
            fc1, fc2 = ffn.layers[0][0], ffn.layers[-2]

            def run_s(rows, w, b, res, post_norm, gather, plan=None, drop_p=0.0, unfolded=False):
                # (unfolded: gather[0] holds EVERY camera's row of a slot, not the first two after a fold launch — the
                # inference kernel only)
                if post_norm is not n0 or (unfolded and (torch.is_grad_enabled() or drop_p > 0
                                                         or not ops.modes().chain_gather_all)):
                    return None
                if torch.is_grad_enabled():         # the same kernel as an autograd Function (train_ops.py)
                    if plan is None or plan.row_query32 is None:
                        return None
                    from .. import train_ops
                    dyn = plan.dynamic
                    fold = (plan.q_rows_all, plan.n_extra_dev) if (dyn and plan.q_rows_all is not None
                                                                   and plan.q_rows_all.shape[1] > 2) else None
                    lay = list(ffn.layers)
                    ph, po = (lay[0][2].p, lay[2].p) if ffn.training else (0.0, 0.0)
                    return train_ops.seam_s(rows, w, b, res, n0, fc1, fc2, n1, gather=gather, row_slot=plan.row_query32,
                                            nrows=plan.nrows_dev if dyn else None, fold=fold, tag="sca_out_ffn_chain",
                                            drop_p=(drop_p, ph, po))
                if drop_p > 0:
                    return None
                if tsa_seam is not None and i + 4 == len(order):
                    # ... and the next layer's TemporalSelfAttention projection of the result behind it
                    done = ops.proj_ffn_chain(rows, w, b, res, n0, fc1, fc2, n1, gather=gather, tag="sca_out_ffn_chain",
                                              tail=(tsa_seam["first"], tsa_seam["pos"], tsa_seam["w"], tsa_seam["b"]))
                    if done is None:
                        return None
                    tsa_seam["proj"] = done[1]
                    return done[0]
                return ops.proj_ffn_chain(rows, w, b, res, n0, fc1, fc2, n1, gather=gather, tag="sca_out_ffn_chain")
            return run_s

        def _chain_t(i):
            """self_attn at ``i`` followed by norm, cross_attn: a callable that runs the output projection,
            "+ identity", the norm AND the cross-attention's merged offset / weight projection in one kernel."""
            if not (fuse_norm or fuse_chain_grad) or tuple(order[i + 1:i + 3]) != ("norm", "cross_attn") \
                    or query_pos is not None or attn_i + 1 >= len(self.attentions):
                return None
            da = getattr(self.attentions[attn_i + 1], "deformable_attention", None)
            n0 = self.norms[norm_i]
            if da is None or not hasattr(da, "sampling_offsets") or not isinstance(n0, torch.nn.LayerNorm):
                return None

            def run(rows, w, b, res, post_norm, drop_p=0.0):
                if post_norm is not n0:
                    return None
                wm, bm = ops.merged_linear_params(da, da.sampling_offsets, da.attention_weights)
                if torch.is_grad_enabled():         # the same kernel as an autograd Function (train_ops.py)
                    from .. import train_ops
                    return train_ops.seam_t(rows, w, b, res, n0, wm, bm, tag="tsa_out_sca_proj_chain", drop_p=drop_p)
                if drop_p > 0:
                    return None
                return ops.proj_ln_proj_chain(rows, w, b, res, n0, wm, bm, tag="tsa_out_sca_proj_chain")
            return run

        skip_norm = False                       # the previous step returned ops.Normed
        skip_ops = 0                            # steps already applied by a chained kernel (ops.Chained)
        next_proj = None                        # the cross-attention's query projection, when a chained kernel made it
        for i, op in enumerate(order):
            if skip_ops:
                skip_ops -= 1
                if op == "norm":
                    norm_i += 1
                elif op == "ffn":
                    ffn_i += 1
                continue
            if op == "self_attn":
                query = self.attentions[attn_i](
                    query, prev_bev, prev_bev, identity if self.pre_norm else None,
                    query_pos=bev_pos, key_pos=bev_pos, attn_mask=attn_masks[attn_i],
                    key_padding_mask=query_key_padding_mask, reference_points=ref_2d,
                    spatial_shapes=bev_shapes, level_start_index=bev_start,
                    defer_residual=_defer(i), post_norm=_post_norm(i), chain=_chain_t(i),
                    offs_attn_proj=tsa_proj if i == 0 else None,
                    bev_hw=(bev_h, bev_w) if (bev_h and bev_w and frame_plan is not None) else None, **kwargs)
                attn_i += 1
                if isinstance(query, ops.NormedWithProj):
                    query, next_proj, skip_norm = query.t, query.proj, True
                elif isinstance(query, ops.Normed):
                    query, skip_norm = query.t, True
                elif isinstance(query, tuple):
                    pending, query = query, None
                else:
                    identity = query
            elif op == "norm":
                norm = self.norms[norm_i]
                if skip_norm:
                    skip_norm = False
                elif pending is not None:
                    branch, res = pending
                    pending = None
                    if torch.is_grad_enabled():
                        query = ops.add_layernorm_autograd(branch, res, norm)
                    else:
                        query = ops.add_layernorm(branch, res, norm.weight, norm.bias, norm.eps)
                    if query is None:
                        query = norm(branch + res)
                else:
                    query = norm(query)
                norm_i += 1
            elif op == "cross_attn":
                query = self.attentions[attn_i](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, reference_points=ref_3d,
                    reference_points_cam=reference_points_cam, mask=mask,
                    attn_mask=attn_masks[attn_i], key_padding_mask=key_padding_mask,
                    spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                    frame_plan=frame_plan, defer_residual=_defer(i), post_norm=_post_norm(i), chain=_chain(i),
                    query_proj=next_proj, **kwargs)
                next_proj = None
                attn_i += 1
                if isinstance(query, ops.Chained):
                    query, skip_ops = query.t, 3
                    identity = query
                elif isinstance(query, ops.Normed):
                    query, skip_norm = query.t, True
                elif isinstance(query, tuple):
                    pending, query = query, None
                else:
                    identity = query
            elif op == "ffn":
                ffn = self.ffns[ffn_i]
                if _defer(i) and isinstance(ffn, FFN):
                    query = ffn(query, identity if self.pre_norm else None, defer_residual=True,
                                post_norm=_post_norm(i))
                else:
                    query = ffn(query, identity if self.pre_norm else None)
                ffn_i += 1
                if isinstance(query, ops.Normed):
                    query, skip_norm = query.t, True
                elif isinstance(query, tuple):
                    pending, query = query, None
        return query
