"""Decoder-side users of the deformable-attention operator.

``CustomMSDeformableAttention`` and ``DetectionTransformerDecoder`` with the registry names,
constructor arguments, parameters and forward contracts of
projects/mmdet3d_plugin/bevformer/modules/decoder.py:53-129 and :133-345.  The attention runs
on the same kernels as the encoder: one merged GEMM for sampling offsets + attention logits, and
softmax / ``reference + offset / (W, H)`` / sampling / aggregation in the fused D = 32 kernel
(one BEV level, 4 points, 900 object queries at base), the unfused operator under autograd
or for box-shaped (4-d) reference points.

The decoder *layer* type of the reference configs (``DetrTransformerDecoderLayer`` with
mmcv's ``MultiheadAttention`` self-attention) is third-party and comes from mmcv / mmdet when
they are installed; the layer sequence here accepts any registered layer type (the tests build
it from this package's ``MyCustomBaseTransformerLayer``).
"""
import warnings

import torch
import torch.nn as nn

from .. import ops
from ..registry import (ATTENTION, HAVE_MMCV, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, BaseModule,
                        constant_, xavier_uniform_)
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer
from .encoder import TransformerLayerSequence
from .temporal_self_attention import _direction_grid, _is_power_of_2


def inverse_sigmoid(x, eps=1e-5):
    """decoder.py:34-50."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


@TRANSFORMER_LAYER_SEQUENCE.register_module(force=True)
class DetectionTransformerDecoder(TransformerLayerSequence):
    """Layer loop with iterative reference-point refinement (decoder.py:53-129)."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.fp16_enabled = False

    def forward(self, query, *args, reference_points=None, reg_branches=None, key_padding_mask=None,
                **kwargs):
        """query (num_query, bs, C); reference_points (bs, num_query, 3) -> (output,
        reference_points), stacked over layers with ``return_intermediate``."""
        output = query
        intermediate, intermediate_reference_points = [], []
        for lid, layer in enumerate(self.layers):
            reference_points_input = reference_points[..., :2].unsqueeze(2)   # (bs, nq, 1 level, 2)
            output = layer(output, *args, reference_points=reference_points_input,
                           key_padding_mask=key_padding_mask, **kwargs)
            output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = reg_branches[lid](output)
                assert reference_points.shape[-1] == 3
                new_reference_points = torch.zeros_like(reference_points)
                new_reference_points[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points[..., :2])
                new_reference_points[..., 2:3] = tmp[..., 4:5] + inverse_sigmoid(reference_points[..., 2:3])
                reference_points = new_reference_points.sigmoid().detach()
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


@ATTENTION.register_module(force=True)
class CustomMSDeformableAttention(BaseModule):
    """Object queries attend to the BEV grid (decoder.py:133-345)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
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
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        constant_(self.sampling_offsets, 0.0)
        self.sampling_offsets.bias.data = _direction_grid(self.num_heads, self.num_levels,
                                                          self.num_points)
        constant_(self.attention_weights, 0.0, 0.0)
        xavier_uniform_(self.value_proj)
        xavier_uniform_(self.output_proj)
        self._is_init = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag="decoder", **kwargs):
        """query (num_query, bs, C) [(bs, num_query, C) with batch_first]; value (num_value,
        bs, C); reference_points (bs, num_query, num_levels, 2 | 4) -> same layout as query."""
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, C = query.shape
        num_value = value.shape[1]
        M, L, P = self.num_heads, self.num_levels, self.num_points

        v = ops.linear_or_torch(value, self.value_proj.weight, self.value_proj.bias,
                                tag="dec_value_proj")
        if key_padding_mask is not None:
            v = v.masked_fill(key_padding_mask[..., None], 0.0)
        v = v.reshape(bs, num_value, M, -1)

        n_off = self.sampling_offsets.out_features
        w, b = ops.merged_linear_params(self, self.sampling_offsets, self.attention_weights)
        proj = ops.linear_or_torch(query.reshape(bs * num_query, C), w, b, tag="dec_offs_attn")
        out = None
        if reference_points.shape[-1] == 2 and ops.fused_wanted(proj, v):
            ref = reference_points.reshape(bs * num_query, 1, L, 2)
            out = ops.msda_fused(v, spatial_shapes, level_start_index, proj, n_off, ref, None, M=M,
                                 L=L, P=P, K=1, off_head=L * P * 2, off_k=0, lg_head=L * P, lg_k=0,
                                 ref_mode=1, vmul=1, vadd=0, Q=num_query, tag="dec_fwd")
            if out is not None:
                out = out.to(query.dtype).view(bs, num_query, C)
        if out is None:
            off = proj[:, :n_off].view(bs, num_query, M, L, P, 2)
            att = proj[:, n_off:].view(bs, num_query, M, L * P).softmax(-1).view(bs, num_query, M, L, P)
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
            out = ops.msda(v, spatial_shapes, level_start_index, loc.contiguous(), att.contiguous(),
                           self.im2col_step, tag="dec_fwd")
        out = ops.linear_or_torch(out, self.output_proj.weight, self.output_proj.bias,
                                  tag="dec_output_proj")
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


# ---------------------------------------------------------------------------
# The two third-party classes the reference's decoder config names
# (bevformer_base.py:110-127): mmcv's ``MultiheadAttention`` wrapper and mmdet's
# ``DetrTransformerDecoderLayer``.  Neither lives in the reference tree (mmcv-full 1.4.0 /
# mmdet 2.14.0, docs/install.md:27,33); when those packages are installed their own classes are
# used and nothing below is registered.  Without them, these restatements — from the published
# behaviour of the two classes, NOT pinned against their source (third-party, absent) — let the
# reference's decoder config build and ``PerceptionTransformer.forward`` run stand-alone.
# ---------------------------------------------------------------------------

class MultiheadAttention(BaseModule):
    """``identity + dropout(proj_drop(nn.MultiheadAttention(q + q_pos, k + k_pos, v)))`` with
    mmcv's defaults: key = query, value = key, identity = query, key_pos = query_pos when the
    shapes agree; parameters under ``attn.*`` (``in_proj_weight``, ``in_proj_bias``,
    ``out_proj.{weight,bias}``)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type="Dropout", drop_prob=0.0), init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__(init_cfg)
        dropout_layer = dict(dropout_layer) if dropout_layer else None
        if "dropout" in kwargs:             # deprecated spelling used by the BEVFormer configs
            attn_drop = kwargs["dropout"]
            if dropout_layer is not None:
                dropout_layer["drop_prob"] = kwargs.pop("dropout")
            else:
                kwargs.pop("dropout")
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        p = dropout_layer.get("drop_prob", 0.0) if dropout_layer else 0.0
        self.dropout_layer = nn.Dropout(p) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None:
            if query_pos.shape == key.shape:
                key_pos = query_pos
            else:
                warnings.warn(f"position encoding of key is missing in {self.__class__.__name__}.")
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class DetrTransformerDecoderLayer(MyCustomBaseTransformerLayer):
    """mmdet's decoder layer: the generic op-order layer with ``batch_first=False`` and the
    six-operation order (self_attn, norm, cross_attn, norm, ffn, norm)."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2,
                 **kwargs):
        kwargs.setdefault("batch_first", False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                             ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                             norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == set(["self_attn", "norm", "cross_attn", "ffn"])


if not HAVE_MMCV:
    ATTENTION.register_module(name="MultiheadAttention", module=MultiheadAttention)
    TRANSFORMER_LAYER.register_module(name="DetrTransformerDecoderLayer", module=DetrTransformerDecoderLayer)
