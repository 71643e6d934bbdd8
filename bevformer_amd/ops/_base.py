"""Shared helpers of the ``ops`` modules: the modes of the calling thread, cache keys, the kernel timers."""
import ctypes
import math
import os

import torch
from torch.autograd.function import Function, once_differentiable

from .. import _lib
from ..ext import _ptr, _req
from ..functions import MultiScaleDeformableAttnFunction_fp32

from .. import modes as _modes


def _forward_modes(backward):
    """Decorator of an autograd Function's ``backward``: runs it under the modes its ``forward`` saw (``ctx.modes``) —
    the autograd engine calls backward on its own thread, outside any ``using`` block of the caller."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        with _modes.activate(ctx.modes):
            return backward(ctx, *grads)
    return wrapped


def _ver(t):
    """Version counter of a tensor for cache keys; inference tensors (``torch.inference_mode``) track none — and cannot
    be written to outside inference mode — so their address alone keys the cache."""
    return 0 if t.is_inference() else t._version


def _m():
    """The modes of this call (bevformer_amd/modes.py): the calling thread's ``using`` block or the process defaults."""
    return _modes.current()


_TIMER = {"cb": None}


def set_kernel_timer(cb):
    """``cb(tag, algorithmic_bytes)`` must return a context manager; it brackets
    every sampling-kernel launch (bench.py records HIP events on the launch
    stream with it).  ``None`` removes the hook."""
    _TIMER["cb"] = cb


class _NoTimer:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _timed(tag, value, loc, attn, out_elems):
    cb = _TIMER["cb"]
    if cb is None:
        return _NoTimer()
    alg = value.numel() * value.element_size() + loc.numel() * 4 + attn.numel() * 4 \
        + out_elems * value.element_size()
    return cb(tag, alg)


def value_storage():
    return _m().value_storage


def modes():
    """The modes of this call (read-only use; edit through ``using`` or the ``set_*`` functions)."""
    return _m()


def using(**overrides):
    """``with ops.using(gemm="bf16", value_storage=torch.bfloat16): ...`` — modes for the calling thread only
    (``bevformer_amd.modes.using``); the ``set_*`` functions below edit the process-wide defaults instead."""
    return _modes.using(**overrides)


def set_value_storage(dtype):
    """fp32 (reference semantics, default) or bf16 storage of the projected
    value tensor inside the sampling kernels (fp32 arithmetic either way)."""
    assert dtype in (torch.float32, torch.bfloat16)
    _modes.process_defaults().value_storage = dtype


_ZERO_SCALARS = {}


def _zero_scalar(dtype, device):
    """A 0-d zero per (dtype, device), made once (outside any stream capture): the zero-stride placeholder gradients of
    tensors whose real gradient travels through a sink."""
    key = (dtype, torch.device(device))
    z = _ZERO_SCALARS.get(key)
    if z is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros((), dtype=dtype, device=device)
        z = _ZERO_SCALARS[key] = torch.zeros((), dtype=dtype, device=device)
    return z
