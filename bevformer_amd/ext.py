"""``_ext``-shaped operator module: the two functions the reference loads with
``ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])``
(projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:10-12),
with mmcv's argument lists (call sites ibid. :118-124 and :150-160), backed by
the hand-written HIP kernels in ``lib/libbevmsda.so``.

Tensors are validated here (device, dtype, contiguity) and raise
``RuntimeError`` on violation, like mmcv's ``AT_ASSERTM`` checks did; then raw
device pointers, sizes and the *current* HIP stream go through the C ABI.
Backward is called from the autograd engine's thread, so the device is guarded
explicitly.
"""
import torch

from . import _lib

_FLOAT = (torch.float32, torch.bfloat16)


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_inputs(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    _req(value.is_cuda, "bevmsda: value must be a GPU tensor (there is no CPU path)")
    dev = value.device
    for name, t in (("value_spatial_shapes", spatial_shapes),
                    ("value_level_start_index", level_start_index),
                    ("sampling_locations", sampling_locations),
                    ("attention_weights", attention_weights)):
        _req(t.device == dev, f"bevmsda: {name} must be on {dev}, got {t.device}")
        _req(t.is_contiguous(), f"bevmsda: {name} must be contiguous")
    _req(value.is_contiguous(), "bevmsda: value must be contiguous")
    _req(value.dtype in _FLOAT, f"bevmsda: unsupported value dtype {value.dtype}")
    _req(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
         "bevmsda: spatial_shapes / level_start_index must be int64")
    _req(sampling_locations.dtype == torch.float32 and attention_weights.dtype == torch.float32,
         "bevmsda: sampling_locations / attention_weights must be float32")
    _req(value.dim() == 4, "bevmsda: value must be (bs, num_keys, num_heads, dim_per_head)")
    _req(sampling_locations.dim() == 6 and sampling_locations.shape[-1] == 2,
         "bevmsda: sampling_locations must be (bs, num_queries, num_heads, num_levels, num_points, 2)")
    N, S, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    _req(tuple(sampling_locations.shape[:3]) == (N, Q, M),
         "bevmsda: sampling_locations does not match value in (bs, ., num_heads)")
    _req(tuple(attention_weights.shape) == (N, Q, M, L, P),
         "bevmsda: attention_weights must be (bs, num_queries, num_heads, num_levels, num_points)")
    _req(tuple(spatial_shapes.shape) == (L, 2) and tuple(level_start_index.shape) == (L,),
         "bevmsda: spatial_shapes must be (num_levels, 2) and level_start_index (num_levels,)")
    return N, S, M, D, L, Q, P


def _ptr(t):
    return t.data_ptr() if t.numel() else None


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                           sampling_locations, attention_weights, im2col_step=64, tuning=None):
    """-> Tensor (bs, num_queries, num_heads*dim_per_head), same dtype as value.

    ``im2col_step`` is accepted for signature compatibility and ignored (the
    HIP kernel has no batch chunking)."""
    N, S, M, D, L, Q, P = _check_inputs(value, value_spatial_shapes, value_level_start_index,
                                        sampling_locations, attention_weights)
    lib = _lib.load()
    out = torch.empty((N, Q, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        args = (_ptr(value), _ptr(value_spatial_shapes), _ptr(value_level_start_index),
                _ptr(sampling_locations), _ptr(attention_weights), N, S, M, D, L, Q, P,
                _ptr(out), stream)
        if value.dtype == torch.float32:
            if tuning is not None:
                rc = lib.bevmsda_forward_f32_ex(*args, tuning)
            else:
                rc = lib.bevmsda_forward_f32(*args)
        elif tuning is not None:
            rc = lib.bevmsda_forward_bf16_ex(*args, tuning)
        else:
            rc = lib.bevmsda_forward_bf16(*args)
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index,
                            sampling_locations, attention_weights, grad_output, grad_value,
                            grad_sampling_loc, grad_attn_weight, im2col_step=64, tuning=None):
    """Fills the three caller-allocated gradient buffers; returns None.

    ``grad_value`` is accumulated into (the reference zeroes it first,
    multi_scale_deformable_attn_function.py:146); ``grad_sampling_loc`` and
    ``grad_attn_weight`` are overwritten.  ``grad_value`` is always float32."""
    N, S, M, D, L, Q, P = _check_inputs(value, value_spatial_shapes, value_level_start_index,
                                        sampling_locations, attention_weights)
    dev = value.device
    _req(grad_output.device == dev and grad_output.is_contiguous()
         and grad_output.dtype == value.dtype and tuple(grad_output.shape) == (N, Q, M * D),
         "bevmsda: grad_output must be a contiguous (bs, num_queries, embed_dims) tensor "
         "of value's dtype")
    for name, t, ref in (("grad_value", grad_value, value),
                         ("grad_sampling_loc", grad_sampling_loc, sampling_locations),
                         ("grad_attn_weight", grad_attn_weight, attention_weights)):
        _req(t.device == dev and t.is_contiguous() and t.dtype == torch.float32
             and t.shape == ref.shape, f"bevmsda: {name} must be a contiguous float32 tensor "
             f"shaped like its forward input")
    lib = _lib.load()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        args = (_ptr(value), _ptr(value_spatial_shapes), _ptr(value_level_start_index),
                _ptr(sampling_locations), _ptr(attention_weights), _ptr(grad_output),
                N, S, M, D, L, Q, P, _ptr(grad_value), _ptr(grad_sampling_loc),
                _ptr(grad_attn_weight), stream)
        if value.dtype == torch.float32:
            if tuning is not None:
                rc = lib.bevmsda_backward_f32_ex(*args, tuning)
            else:
                rc = lib.bevmsda_backward_f32(*args)
        elif tuning is not None:
            rc = lib.bevmsda_backward_bf16_ex(*args, tuning)
        else:
            rc = lib.bevmsda_backward_bf16(*args)
    _lib.check(rc, "ms_deform_attn_backward")
    return None
