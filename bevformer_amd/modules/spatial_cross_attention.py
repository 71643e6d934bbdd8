"""SpatialCrossAttention / MSDeformableAttention3D on the MI355X kernels.

Registry names, constructor arguments, parameter names and forward contracts
follow the reference (projects/mmdet3d_plugin/bevformer/modules/
spatial_cross_attention.py:32-175 and :179-399).  The execution differs:

  * no zero-padded (bs, num_cams, max_len) rebatch and no per-layer
    ``nonzero()`` host syncs: the visible (camera, query) pairs are a ragged row
    list built once per frame (``modules/geometry.py``) and the sampling kernel
    reads the value of the right camera through a row->batch table.  Padded rows
    in the reference are computed and never read back (:165-167), so outputs
    are identical;
  * offsets and attention logits come from ONE GEMM over the concatenated
    projection weights;
  * the per-camera scatter / camera-count division is an ``index_add_`` over the
    row list and a multiply by the precomputed reciprocal count.
"""
import warnings

import torch
import torch.nn as nn

from .. import ops
from ..registry import ATTENTION, BaseModule, build_attention, constant_, force_fp32, xavier_uniform_
from . import geometry
from .temporal_self_attention import _direction_grid, _is_power_of_2


@ATTENTION.register_module(force=True)
class MSDeformableAttention3D(BaseModule):
    """Deformable attention over the Z-anchors of a BEV pillar
    (spatial_cross_attention.py:179-399); ``output_proj`` is ``None`` as in the
    reference (:221) — the projection lives in SpatialCrossAttention."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, "
                             f"but got {embed_dims} and {num_heads}")
        if not _is_power_of_2(embed_dims // num_heads):
            warnings.warn("You'd better set embed_dims in MultiScaleDeformAttention to make the "
                          "dimension of each attention head a power of 2 (the HIP kernel's "
                          "16-byte lane-group path needs a multiple of 4).")
        self.norm_cfg = norm_cfg
        self.batch_first = batch_first
        self.output_proj = None
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        constant_(self.sampling_offsets, 0.0)
        self.sampling_offsets.bias.data = _direction_grid(self.num_heads, self.num_levels,
                                                          self.num_points)
        constant_(self.attention_weights, 0.0, 0.0)
        xavier_uniform_(self.value_proj)
        xavier_uniform_(self.output_proj)      # None: skipped, as in mmcv
        self._is_init = True

    # -- shared pieces -----------------------------------------------------
    def _project_queries(self, query):
        """(..., C) -> offsets (..., M, L, P, 2), softmaxed weights (..., M, L, P)
        from one GEMM over [sampling_offsets ; attention_weights]."""
        M, L, P = self.num_heads, self.num_levels, self.num_points
        w, b = ops.merged_linear_params(self, self.sampling_offsets, self.attention_weights)
        return self._split_projection(ops.linear_or_torch(query, w, b, tag="sca_offs_attn"))

    def _split_projection(self, proj):
        M, L, P = self.num_heads, self.num_levels, self.num_points
        n_off = self.sampling_offsets.out_features
        lead = proj.shape[:-1]
        off = proj[..., :n_off].reshape(*lead, M, L, P, 2)
        att = proj[..., n_off:].reshape(*lead, M, L * P).softmax(-1).view(*lead, M, L, P)
        return off, att

    def _locations(self, off, reference_points, spatial_shapes):
        """Z-anchor sampling locations (spatial_cross_attention.py:357-372):
        point k of a level uses anchor k mod num_Z_anchors."""
        if reference_points.shape[-1] != 2:
            if reference_points.shape[-1] == 4:
                assert False
            raise ValueError(f"Last dim of reference_points must be 2 or 4, "
                             f"but get {reference_points.shape[-1]} instead.")
        M, L, P = self.num_heads, self.num_levels, self.num_points
        lead = off.shape[:-4]
        Dz = reference_points.shape[-2]
        assert P % Dz == 0
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        off = off / normalizer[:, None, :]                       # (...,M,L,P,2) / (L,1,2)
        off = off.view(*lead, M, L, P // Dz, Dz, 2)
        ref = reference_points.view(*lead, 1, 1, 1, Dz, 2)
        return (ref + off).view(*lead, M, L, P, 2)

    # -- reference-shaped call (batch layout) --------------------------------
    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):
        """query (bs, Q, C), value (bs, S, C), reference_points (bs, Q, Dz, 2)
        -> (bs, Q, C).  No residual, no dropout, no output projection here."""
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, Q, _ = query.shape
        num_value = value.shape[1]
        v = ops.linear_or_torch(value, self.value_proj.weight, self.value_proj.bias,
                                tag="sca_value_proj")
        if key_padding_mask is not None:
            v = v.masked_fill(key_padding_mask[..., None], 0.0)
        v = v.view(bs, num_value, self.num_heads, -1)
        off, att = self._project_queries(query)
        loc = self._locations(off, reference_points, spatial_shapes)
        out = ops.msda(v, spatial_shapes, level_start_index, loc.contiguous(), att.contiguous(),
                       self.im2col_step)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return out

    # -- ragged call used by SpatialCrossAttention ---------------------------
    def project_value(self, value, shared=None):
        """(N, S, C) camera features -> (N, S, M, D).  ``shared``: the ``ops.GradThread`` of all layers' projections of
        these features."""
        v = ops.linear_or_torch(value, self.value_proj.weight, self.value_proj.bias,
                                tag="sca_value_proj", thread=shared)
        return v.view(value.shape[0], value.shape[1], self.num_heads, -1)

    def forward_rows_shared_projection(self, queries, value, row_ref, row_batch, row_src,
                                       spatial_shapes, level_start_index, frame_plan=None, autograd=False,
                                       q_rows=None, proj=None):
        """queries (Q, C) projected once; row r samples with the projection row
        ``row_src[r]`` and its own anchors ``row_ref[r]`` -> (R, C), or None when
        the fused kernel does not cover the shape.  ``autograd``: through
        ``ops.msda_fused_autograd`` (gradients w.r.t. the value and the projection rows)."""
        M, L, P = self.num_heads, self.num_levels, self.num_points
        Dz = row_ref.shape[-2]
        if P % Dz != 0 or row_ref.shape[-1] != 2:
            return None
        n_off = self.sampling_offsets.out_features
        if proj is None:            # (else: the merged projection of these rows came out of the previous step's kernel)
            w, b = ops.merged_linear_params(self, self.sampling_offsets, self.attention_weights)
            proj = ops.linear_or_torch(queries, w, b, tag="sca_offs_attn")
        if autograd:
            if value.shape[-1] != 32 or L > 4 or P not in (4, 8) or value.dtype != torch.float32:   # (storage may be bf16)
                return None
            return ops.msda_fused_autograd(value, spatial_shapes, level_start_index, proj, n_off,
                                           row_ref.reshape(-1, 1, Dz, 2), row_batch, M=M, L=L, P=P, K=1,
                                           off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0,
                                           vmul=1, vadd=0, row_src=row_src, q_rows=q_rows,
                                           tag="sca_fwd").to(queries.dtype)
        lds = {}
        if frame_plan is not None and frame_plan.dynamic:
            # row count on the device (geometry.DevicePlanner); launch sized by the planner's hint
            lds = dict(nrows=frame_plan.nrows_dev, launch_rows=frame_plan.launch_rows)
        out = ops.msda_fused(value, spatial_shapes, level_start_index, proj, n_off,
                             row_ref.reshape(-1, 1, Dz, 2), row_batch, M=M, L=L, P=P, K=1,
                             off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0,
                             vmul=1, vadd=0, row_src=row_src, tag="sca_fwd", **lds)
        return None if out is None else out.to(queries.dtype)

    def forward_ragged(self, query_rows, value, row_ref, row_batch, spatial_shapes,
                       level_start_index):
        """query_rows (R, C); value (N, S, M, D) already projected; row_ref
        (R, Dz, 2); row_batch (R,) int32 -> (R, C)."""
        M, L, P = self.num_heads, self.num_levels, self.num_points
        Dz = row_ref.shape[-2]
        n_off = self.sampling_offsets.out_features
        w, b = ops.merged_linear_params(self, self.sampling_offsets, self.attention_weights)
        proj = ops.linear_or_torch(query_rows, w, b, tag="sca_offs_attn")
        if P % Dz == 0 and row_ref.shape[-1] == 2 and ops.fused_wanted(proj, value):
            # raw projection row -> softmax / locations / sampling in ONE kernel
            out = ops.msda_fused(value, spatial_shapes, level_start_index, proj, n_off,
                                 row_ref.reshape(-1, 1, Dz, 2), row_batch, M=M, L=L, P=P, K=1,
                                 off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0, ref_mode=0,
                                 vmul=1, vadd=0, tag="sca_fwd")
            if out is not None:
                return out.to(query_rows.dtype)
        off, att = self._split_projection(proj)
        loc = self._locations(off, row_ref, spatial_shapes)
        return ops.msda_ragged(value, spatial_shapes, level_start_index, loc.contiguous(),
                               att.contiguous(), row_batch, tag="sca_fwd")


@ATTENTION.register_module(force=True)
class SpatialCrossAttention(BaseModule):
    """BEV queries attend to the camera features they project into
    (spatial_cross_attention.py:32-175)."""

    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False,
                 deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=256,
                                           num_levels=4),
                 **kwargs):
        super().__init__(init_cfg)
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.init_weight()

    def init_weight(self):
        xavier_uniform_(self.output_proj)

    @force_fp32(apply_to=("query", "key", "value", "query_pos", "reference_points_cam"))   # spatial_cross_attention.py:75
    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, reference_points_cam=None,
                bev_mask=None, level_start_index=None, flag="encoder", frame_plan=None,
                projected_value=None, defer_residual=False, post_norm=None, chain=None, query_proj=None, **kwargs):
        """query (bs, Q, C); key/value (Nc, S, bs, C); reference_points_cam
        (Nc, bs, Q, Dz, 2); bev_mask (Nc, bs, Q, Dz) -> (bs, Q, C).

        ``frame_plan`` (from this package's encoder) carries the ragged row list;
        without it the list is derived from ``bev_mask`` here (one host sync,
        like the reference's ``nonzero()``)."""
        if key is None:
            key = query
        if value is None:
            value = key
        if residual is None:
            inp_residual = query
        else:  # the reference leaves `slots` undefined on this path (:128-130)
            inp_residual = residual
        if query_pos is not None:
            query = query + query_pos
        bs, Q, C = query.shape

        if frame_plan is not None:
            row_query, row_batch, row_ref, inv_count = (frame_plan.row_query, frame_plan.row_batch,
                                                        frame_plan.row_ref, frame_plan.inv_count)
        else:
            row_query, row_batch, row_ref, inv_count, _ = geometry.build_sca_rows(
                reference_points_cam, bev_mask)

        ready = kwargs.pop("projected_value_ready", None)
        if ready is not None:           # the hoisted projection ran on a side stream: join it here
            torch.cuda.current_stream(query.device).wait_event(ready)
        if projected_value is None:
            Nc, S, _, _ = value.shape
            feats = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, S, self.embed_dims)
            share = kwargs.get("value_grad_share")
            projected_value = self.deformable_attention.project_value(feats, None if share is None else share.get("sca"))

        da = self.deformable_attention
        slots, projected = None, False
        if frame_plan is not None and frame_plan.q_rows is not None \
                and ops.fused_wanted(query, projected_value):
            # inference path: every BEV query is projected ONCE (the reference projects a
            # copy per camera, spatial_cross_attention.py:149-153); each (camera, query) row
            # reads its query's projection row through row_src; the camera mean is a gather
            out_rows = da.forward_rows_shared_projection(
                query.reshape(bs * Q, C), projected_value, row_ref, row_batch,
                frame_plan.row_query32, spatial_shapes, level_start_index, frame_plan=frame_plan,
                proj=query_proj if query_pos is None else None)
            if out_rows is not None:
                # slots seen by more than two cameras (rare; known to the device only): the chain kernel walks every
                # camera's row of a slot itself (idx = q_rows_all); any other consumer gathers TWO rows per slot, and a
                # stand-alone launch folds the third.. rows into the first beforehand (a no-op launch on most frames)
                extra = frame_plan.dynamic and frame_plan.q_rows_all is not None and frame_plan.q_rows_all.shape[1] > 2
                chain_ok = post_norm is not None and chain is not None and not (self.training and self.dropout.p > 0)
                if chain_ok and extra:
                    done = chain(out_rows, self.output_proj.weight, self.output_proj.bias, inp_residual, post_norm,
                                 (frame_plan.q_rows_all, inv_count), frame_plan, unfolded=True)
                    if done is not None:
                        return ops.Chained(done.view(bs, Q, C))
                if frame_plan.dynamic:
                    ops.fold_extra_rows(out_rows, frame_plan.q_rows_all, frame_plan.n_extra_dev)
                if chain_ok:
                    # ... and the FFN and its norm behind them: the whole row-local tail of the layer in one kernel
                    done = chain(out_rows, self.output_proj.weight, self.output_proj.bias, inp_residual, post_norm,
                                 (frame_plan.q_rows, inv_count), frame_plan)
                    if done is not None:
                        return ops.Chained(done.view(bs, Q, C))
                if post_norm is not None and not (self.training and self.dropout.p > 0):
                    # camera mean + output projection + "+ residual" + the layer's norm in one kernel
                    fused = ops.linear_layernorm(out_rows, self.output_proj.weight, self.output_proj.bias,
                                                 inp_residual, post_norm, gather=(frame_plan.q_rows, inv_count),
                                                 tag="sca_output_proj")
                    if fused is not None:
                        return ops.Normed(fused.view(bs, Q, C))
                # camera mean + output projection in one kernel where the GEMM kernel is in use
                proj = ops.linear_gather_mean(out_rows, frame_plan.q_rows, inv_count,
                                              self.output_proj.weight, self.output_proj.bias,
                                              tag="sca_output_proj")
                if proj is not None:
                    slots, projected = proj.view(bs, Q, C), True
                else:
                    slots = ops.gather_mean(out_rows, frame_plan.q_rows, inv_count).view(bs, Q, C)
        if slots is None and chain is not None and post_norm is not None and torch.is_grad_enabled() \
                and frame_plan is not None and frame_plan.q_rows is not None and frame_plan.q_rows.shape[1] == 2 \
                and frame_plan.row_query32 is not None and query_pos is None and bs * Q == frame_plan.q_rows.shape[0] \
                and ops.fused_training_wanted(query, projected_value) \
                and not getattr(projected_value, "_bevmsda_partial", False):
            # autograd fast path (train_ops.py): fused sampling (every query projected once; the projection rows may come
            # out of the previous seam's kernel) -> the whole row-local tail of the layer in one kernel that saves
            # what its backward needs; the ragged row count never leaves the device
            dyn = frame_plan.dynamic
            sink = kwargs.get("projected_value_sink")
            da_ok = projected_value.shape[-1] == 32 and da.num_levels <= 4 and da.num_points in (4, 8) \
                and (projected_value.dtype == torch.float32
                     or (projected_value.dtype == torch.bfloat16 == ops.value_storage() and sink is not None)) \
                and da.num_points % row_ref.shape[-2] == 0 \
                and (not dyn or da.num_levels in (1, 2, 4))     # (device-side row count: the gather-form chain pass)
            q_tab = frame_plan.q_rows_all if (dyn and frame_plan.q_rows_all is not None) else frame_plan.q_rows
            if da_ok and (dyn or frame_plan.row_query32.numel() > 0):
                proj_rows = query_proj
                if proj_rows is None:
                    wm, bm = ops.merged_linear_params(da, da.sampling_offsets, da.attention_weights)
                    proj_rows = ops.linear_or_torch(query.reshape(bs * Q, C), wm, bm, tag="sca_offs_attn")
                M_, L_, P_ = da.num_heads, da.num_levels, da.num_points
                Dz = row_ref.shape[-2]
                out_rows = ops.msda_fused_autograd(
                    projected_value, spatial_shapes, level_start_index, proj_rows, da.sampling_offsets.out_features,
                    row_ref.reshape(-1, 1, Dz, 2), row_batch, M=M_, L=L_, P=P_, K=1, off_head=L_ * P_ * 2, off_k=0,
                    lg_head=L_ * P_, lg_k=0, ref_mode=0, vmul=1, vadd=0, row_src=frame_plan.row_query32, q_rows=q_tab,
                    tag="sca_fwd", nrows=frame_plan.nrows_dev if dyn else None,
                    launch_rows=frame_plan.launch_rows if dyn else 0,
                    value_sink=sink,
                    n_extra=frame_plan.n_extra_dev if dyn else None)
                done = chain(out_rows, self.output_proj.weight, self.output_proj.bias, inp_residual, post_norm,
                             (frame_plan.q_rows, inv_count), frame_plan,
                             drop_p=self.dropout.p if self.training else 0.0)
                if done is not None:
                    return ops.Chained(done.view(bs, Q, C))
                # (declined: the per-op statements below redo the sampling — correct, slower)
        if slots is None:
            if getattr(projected_value, "_bevmsda_partial", False):
                # a hoisted projection that skipped the cameras this rank's queries cannot see (uninitialised rows)
                # is only good for the fused kernel above: project the features here for every other path
                Nc, S, _, _ = value.shape
                feats = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, S, self.embed_dims)
                projected_value = self.deformable_attention.project_value(feats)
            if frame_plan is not None and frame_plan.dynamic:
                # the unfused statements need the row count on the host: one read of the plan's counters
                fp = frame_plan.materialize()
                row_query, row_batch, row_ref = fp.row_query, fp.row_batch, fp.row_ref
            out_rows = None
            if ops.fused_training_wanted(query, projected_value) and row_query.numel() > 0:
                # autograd path: the queries are projected once here too and the fused kernel reads a row's
                # projection through row_src; its backward accumulates over the cameras of a query
                out_rows = da.forward_rows_shared_projection(
                    query.reshape(bs * Q, C), projected_value, row_ref, row_batch,
                    row_query.to(torch.int32), spatial_shapes, level_start_index, autograd=True,
                    q_rows=frame_plan.q_rows if frame_plan is not None else None)
            if out_rows is None:
                q_rows = query.reshape(bs * Q, C).index_select(0, row_query)
                out_rows = da.forward_ragged(q_rows, projected_value, row_ref, row_batch,
                                             spatial_shapes, level_start_index)
            qr = frame_plan.q_rows if frame_plan is not None and not frame_plan.dynamic else None
            if qr is not None and out_rows.is_cuda and out_rows.dtype == torch.float32 and qr.shape[0] == bs * Q \
                    and ops.modes().fused_train and C % 4 == 0 and out_rows.shape[0] > 0:
                # camera sum + division by the camera count as one gather kernel (its backward is a gather too)
                slots = ops.gather_mean_autograd(out_rows, qr, inv_count, row_query).view(bs, Q, C)
            else:
                slots = torch.zeros(bs * Q, C, dtype=out_rows.dtype, device=query.device)
                slots.index_add_(0, row_query, out_rows)
                slots = slots.view(bs, Q, C) * inv_count.to(slots.dtype)
        if not projected:
            slots = ops.linear_or_torch(slots, self.output_proj.weight, self.output_proj.bias,
                                        tag="sca_output_proj")
        if defer_residual:
            return self.dropout(slots), inp_residual    # the layer fuses "+ residual" into its LayerNorm
        return self.dropout(slots) + inp_residual
