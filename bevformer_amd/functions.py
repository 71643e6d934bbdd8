"""Autograd wrapper with the reference's ``.apply`` signature.

Mirrors ``MultiScaleDeformableAttnFunction_fp32``
(projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:90-163):
positional arguments ``(value, value_spatial_shapes, value_level_start_index,
sampling_locations, attention_weights, im2col_step)``; gradients are returned
for positions 0, 3 and 4 only.  Like the reference (``custom_fwd(cast_inputs=
torch.float32)``, ibid. :93) every floating input is computed in fp32; the
``_bf16`` sibling keeps ``value`` in bf16 storage (HBM bytes halved) with fp32
sampling arithmetic — it has no counterpart in the reference.
"""
import torch
from torch.autograd.function import Function, once_differentiable

from . import ext as ext_module


class MultiScaleDeformableAttnFunction_fp32(Function):

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step=64):
        ctx.im2col_step = im2col_step
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        value = value.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        output = ext_module.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations,
            attention_weights, im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        grad_value = torch.zeros_like(value)
        grad_loc = torch.empty_like(loc)
        grad_attn = torch.empty_like(attn)
        ext_module.ms_deform_attn_backward(
            value, shapes, start, loc, attn, grad_output.float().contiguous(),
            grad_value, grad_loc, grad_attn, im2col_step=ctx.im2col_step)
        dv, dl, da = ctx.in_dtypes
        return grad_value.to(dv), None, None, grad_loc.to(dl), grad_attn.to(da), None


class MultiScaleDeformableAttnFunction_bf16(Function):
    """bf16 storage for ``value`` / output, fp32 locations, weights and math."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step=64):
        ctx.im2col_step = im2col_step
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        from . import modes as _modes
        ctx.lanes8 = bool(_modes.current().bf16_lanes8)    # (snapshot: backward runs on the autograd thread)
        value = value.to(torch.bfloat16).contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        output = ext_module.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations,
            attention_weights, im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        grad_value = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
        grad_loc = torch.empty_like(loc)
        grad_attn = torch.empty_like(attn)
        tuning = None
        if ctx.lanes8:                     # benchmark knob (modes.bf16_lanes8): the 8-byte-lane gather kernel
            from . import _lib
            tuning = _lib.Tuning()
            tuning.reserved[3] = 1
        ext_module.ms_deform_attn_backward(
            value, shapes, start, loc, attn, grad_output.to(torch.bfloat16).contiguous(),
            grad_value, grad_loc, grad_attn, im2col_step=ctx.im2col_step, tuning=tuning)
        dv, dl, da = ctx.in_dtypes
        return grad_value.to(dv), None, None, grad_loc.to(dl), grad_attn.to(da), None


# the reference also exports an ``_fp16`` name that is never selected
# (spatial_cross_attention.py:386-389 picks ``_fp32`` in both branches)
MultiScaleDeformableAttnFunction_fp16 = MultiScaleDeformableAttnFunction_fp32
