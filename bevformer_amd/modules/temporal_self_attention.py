"""TemporalSelfAttention on the MI355X deformable-attention kernels.

Same registry name, constructor arguments, parameters (``sampling_offsets``,
``attention_weights``, ``value_proj``, ``output_proj``) and forward contract as
the reference class (projects/mmdet3d_plugin/bevformer/modules/
temporal_self_attention.py:26-272); what differs is how the work is issued:

  * the sampling-offset and attention-weight projections share their input
    (``cat([prev_or_cur, query + pos])``), so they run as ONE GEMM over the
    concatenated weight (same numbers, one launch, input read once);
  * when there is no history (``value is None``) the reference stacks the query
    twice and projects both copies (temporal_self_attention.py:177-180,198);
    here the projection runs once and both queue entries sample the same
    projected tensor through the ragged row->batch table of the kernel;
  * sampling runs in the hand-written HIP kernel (no CPU fallback: on a CPU
    tensor this raises).
"""
import math
import warnings

import torch
import torch.nn as nn

from .. import ops
from ..registry import ATTENTION, BaseModule, constant_, xavier_uniform_


def _rows_layout(reference_points, bs, nq, Q, L):
    """(bs*nq, Q, L, 2) -> (bs*Q, nq, L, 2) contiguous, the layout the fused kernel reads.  The encoder hands every
    layer the same reference tensor: the permuted copy is made once per tensor (keyed on the object and its
    version) instead of once per layer."""
    if reference_points.is_inference():     # (torch.inference_mode: no version counter to key the cache on)
        return reference_points.reshape(bs, nq, Q, L, 2).permute(0, 2, 1, 3, 4).reshape(bs * Q, nq, L, 2).contiguous()
    hit = getattr(reference_points, "_bevmsda_rows", None)
    if hit is not None and hit[0] == (reference_points._version, bs, nq, Q, L):
        return hit[1]
    rows = reference_points.reshape(bs, nq, Q, L, 2).permute(0, 2, 1, 3, 4).reshape(bs * Q, nq, L, 2).contiguous()
    try:
        reference_points._bevmsda_rows = ((reference_points._version, bs, nq, Q, L), rows)
    except AttributeError:
        pass
    return rows


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return (n & (n - 1) == 0) and n != 0


def _direction_grid(num_heads, copies, num_points):
    """Initial sampling-offset bias: one direction per head, radius i+1 for
    point i (temporal_self_attention.py:110-122 / spatial_cross_attention.py:256-267)."""
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(num_heads, 1, 1, 2)
    grid = grid.repeat(1, copies, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    return grid.view(-1)


@ATTENTION.register_module(force=True)
class TemporalSelfAttention(BaseModule):

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_bev_queue=2,
                 im2col_step=64, dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, "
                             f"but got {embed_dims} and {num_heads}")
        if not _is_power_of_2(embed_dims // num_heads):
            warnings.warn("You'd better set embed_dims in MultiScaleDeformAttention to make the "
                          "dimension of each attention head a power of 2 (the HIP kernel's "
                          "16-byte lane-group path needs a multiple of 4).")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_bev_queue = num_bev_queue
        self.sampling_offsets = nn.Linear(embed_dims * num_bev_queue,
                                          num_bev_queue * num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims * num_bev_queue,
                                           num_bev_queue * num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        constant_(self.sampling_offsets, 0.0)
        self.sampling_offsets.bias.data = _direction_grid(
            self.num_heads, self.num_levels * self.num_bev_queue, self.num_points)
        constant_(self.attention_weights, 0.0, 0.0)
        xavier_uniform_(self.value_proj)
        xavier_uniform_(self.output_proj)
        self._is_init = True

    def _sample_unfused(self, proj, n_off, v, reference_points, spatial_shapes, level_start_index,
                        shared_value, bs, Q, C):
        """softmax / location arithmetic as separate torch ops + the operator with
        the reference's signature (the autograd path)."""
        nq, M, L, P = self.num_bev_queue, self.num_heads, self.num_levels, self.num_points
        off = proj[..., :n_off].reshape(bs, Q, M, nq, L, P, 2)
        att = proj[..., n_off:].reshape(bs, Q, M, nq, L * P).softmax(-1)
        att = att.view(bs, Q, M, nq, L, P).permute(0, 3, 1, 2, 4, 5).reshape(bs * nq, Q, M, L, P)
        off = off.permute(0, 3, 1, 2, 4, 5, 6).reshape(bs * nq, Q, M, L, P, 2)

        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            loc = reference_points[:, :, None, :, None, :] \
                + off / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] \
                + off / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, "
                             f"but get {reference_points.shape[-1]} instead.")

        if shared_value:
            # queue entries b*2+0 and b*2+1 both sample projected batch entry b
            row_batch = torch.arange(bs, device=proj.device, dtype=torch.int32) \
                .repeat_interleave(nq * Q)
            out = ops.msda_ragged(v, spatial_shapes, level_start_index,
                                  loc.reshape(bs * nq * Q, M, L, P, 2),
                                  att.reshape(bs * nq * Q, M, L, P), row_batch, tag="tsa_fwd")
            out = out.view(bs * nq, Q, C)
        else:
            out = ops.msda(v, spatial_shapes, level_start_index, loc, att.contiguous(),
                           self.im2col_step, tag="tsa_fwd")
        # mean over the queue entries (temporal_self_attention.py:257-262)
        return out.view(bs, nq, Q, C).mean(1)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag="decoder", bev_slice=None, defer_residual=False,
                tsa_projected_value=None, post_norm=None, chain=None, offs_attn_proj=None, **kwargs):
        """query (bs, Q, C) [batch_first]; value None or (bs*2, Q, C) with index
        b*2+queue; reference_points (bs*2, Q, num_levels, 2) -> (bs, Q, C).

        ``bev_slice=(q0, q1)`` (BEV tiling): ``query`` holds only queries
        [q0, q1) while ``value`` is the full grid.  ``tsa_projected_value``
        (bs*2, Q, M, D): ``value_proj(value)`` already computed by the encoder
        for all layers at once (``hoisted_value_projections``).  ``offs_attn_proj`` (bs*Q, offsets + weights):
        the merged ``sampling_offsets`` / ``attention_weights`` projection of ``cat([value[:bs], query + query_pos], -1)``
        already made by the kernel that produced ``query`` (``BEVFormerEncoder.tsa_seam``)."""
        assert self.num_bev_queue == 2
        shared_value = value is None
        if shared_value:
            assert self.batch_first
        if identity is None:
            identity = query
        query_in = query
        if not self.batch_first:
            # (the reference adds query_pos before the permute; same numbers)
            query = (query + query_pos if query_pos is not None else query).permute(1, 0, 2)
            if value is not None:
                value = value.permute(1, 0, 2)
        bs, Q, C = query.shape
        nq, M, L, P = self.num_bev_queue, self.num_heads, self.num_levels, self.num_points
        if shared_value:
            # the reference's stack([query, query], 1).reshape(bs*2, ...)[:bs] (:180,197):
            # entry i of the stacked tensor is query[i // 2]
            first = query_in if bs == 1 else query_in[torch.arange(bs, device=query.device) // 2]
        else:
            first = value[:bs]
            hist = kwargs.get("tsa_history")
            tiled = kwargs.get("bev_rows") is not None or bev_slice is not None
            if hist is not None and bs == 1 and tiled and tuple(hist.shape) == (bs, Q, first.shape[-1]):
                # BEV tiling: the history rows of MY queries, gathered once per frame by the schedule
                first, tiled = hist, False
            elif hist is not None and bs == 1 and not tiled and hist.shape == first.shape:
                # value[:1] IS the history BEV (stack([prev_bev, bev_query], 1)[0]); taken from the tensor it was stacked
                # from, it needs no gradient when the history is detached (the reference computes it under no_grad,
                # bevformer.py:158-177) — through the stacked tensor autograd computes one, pads it and adds it, per layer
                first = hist
            if tiled and kwargs.get("bev_rows") is not None:              # (BEV tiling, sector layout: my queries' cells)
                first = first.index_select(1, kwargs["bev_rows"])
            elif tiled and bev_slice is not None:
                first = first[:, bev_slice[0]:bev_slice[1]]
        src = query_in if shared_value else value
        num_value = src.shape[1]
        if tsa_projected_value is not None and not shared_value and key_padding_mask is None:
            v = tsa_projected_value
        else:
            share = kwargs.get("value_grad_share") if not shared_value else None
            v = ops.linear_or_torch(src, self.value_proj.weight, self.value_proj.bias,
                                    tag="tsa_value_proj", thread=None if share is None else share.get("tsa"))
            if key_padding_mask is not None:
                v = v.masked_fill(key_padding_mask[..., None], 0.0)
            v = v.reshape(v.shape[0], num_value, M, -1)

        # one GEMM for offsets (nq*M*L*P*2 columns) and weights (nq*M*L*P columns); on the
        # MFMA kernel the cat([first, query + pos]) input is read in place from its two sources
        n_off = self.sampling_offsets.out_features
        w, b = ops.merged_linear_params(self, self.sampling_offsets, self.attention_weights)
        proj = None
        if offs_attn_proj is not None and self.batch_first and not shared_value and key_padding_mask is None \
                and offs_attn_proj.numel() == bs * Q * w.shape[0]:
            proj = offs_attn_proj.view(bs, Q, w.shape[0])
        if proj is None and self.batch_first:
            proj = ops.linear(first, w, b, x2=query_in, x2_add=query_pos, tag="tsa_offs_attn")
            if proj is None and chain is not None and torch.is_grad_enabled() and first.shape == query_in.shape \
                    and first.shape[-1] == 256 and (query_pos is None or query_pos.shape == query_in.shape):
                # autograd fast path (the layer offered its chain kernel): the two-source GEMM as an autograd Function
                from .. import train_ops
                if train_ops.wanted(first, query_in, query_pos, w, b):
                    proj = train_ops.two_source_linear(first, query_in, query_pos, w, b, tag="tsa_offs_attn")
        if proj is None:
            if self.batch_first and query_pos is not None:
                query = query + query_pos
            proj = ops.linear_or_torch(torch.cat([first, query], -1), w, b, tag="tsa_offs_attn")
        out = None
        if reference_points.shape[-1] == 2 and self.batch_first and key_padding_mask is None \
                and ops.fused_wanted(proj, v):
            # softmax, locations, sampling of both queue entries and their mean in ONE kernel
            ref = _rows_layout(reference_points, bs, nq, Q, L)
            out = ops.msda_fused(v, spatial_shapes, level_start_index, proj.view(bs * Q, -1), n_off,
                                 ref, None, M=M, L=L, P=P, K=nq, off_head=nq * L * P * 2,
                                 off_k=L * P * 2, lg_head=nq * L * P, lg_k=L * P, ref_mode=1,
                                 vmul=1 if shared_value else nq, vadd=0 if shared_value else 1,
                                 Q=Q, tag="tsa_fwd",
                                 grid_hw=kwargs.get("bev_hw") if (bs == 1 and not shared_value and num_value == Q) else None)
            if out is not None:
                out = out.to(query.dtype).view(bs, Q, C)
        vsink = kwargs.get("tsa_projected_value_sink") if v is tsa_projected_value else None
        v_ok = v.dtype == torch.float32 or (v.dtype == torch.bfloat16 == ops.value_storage() and vsink is not None)
        if out is None and reference_points.shape[-1] == 2 and self.batch_first and key_padding_mask is None \
                and v.shape[-1] == 32 and L <= 4 and P in (4, 8) and nq * P <= 8 and v_ok \
                and ops.fused_training_wanted(proj, v):
            # autograd path: same kernel, gradients w.r.t. the value and the projection rows
            ref = _rows_layout(reference_points, bs, nq, Q, L)
            out = ops.msda_fused_autograd(v.contiguous(), spatial_shapes, level_start_index,
                                          proj.reshape(bs * Q, -1), n_off, ref, None, M=M, L=L, P=P, K=nq,
                                          off_head=nq * L * P * 2, off_k=L * P * 2, lg_head=nq * L * P,
                                          lg_k=L * P, ref_mode=1, vmul=1 if shared_value else nq,
                                          vadd=0 if shared_value else 1, Q=Q, tag="tsa_fwd",
                                          value_sink=vsink)
            out = out.to(query.dtype).view(bs, Q, C)
        if out is None:
            out = self._sample_unfused(proj, n_off, v, reference_points, spatial_shapes,
                                       level_start_index, shared_value, bs, Q, C)
        drop_p = self.dropout.p if self.training else 0.0
        if post_norm is not None and chain is not None and self.batch_first and (drop_p == 0 or torch.is_grad_enabled()):
            # ... and the next attention's projection of the normed rows behind them, in the same kernel (under autograd
            # with the dropout of train() mode applied inside it)
            done = chain(out, self.output_proj.weight, self.output_proj.bias, identity, post_norm, drop_p=drop_p)
            if done is not None:
                return ops.NormedWithProj(done[0], done[1])
        if post_norm is not None and self.batch_first and not (self.training and self.dropout.p > 0):
            # output_proj, "+ identity" and the layer's norm in one kernel
            fused = ops.linear_layernorm(out, self.output_proj.weight, self.output_proj.bias, identity, post_norm,
                                         tag="tsa_output_proj")
            if fused is not None:
                return ops.Normed(fused)
        out = ops.linear_or_torch(out, self.output_proj.weight, self.output_proj.bias,
                                  tag="tsa_output_proj")
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        if defer_residual:
            return self.dropout(out), identity  # the layer fuses "+ identity" into its LayerNorm
        return self.dropout(out) + identity
