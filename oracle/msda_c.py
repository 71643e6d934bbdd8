"""TEST INFRASTRUCTURE — builds and binds oracle/msda_ref.c (gcc, plain C).

``build()`` compiles ``oracle/_build/libmsda_ref.so``; ``forward``/``backward``
take CPU torch tensors in the operator's layout and return CPU tensors.
Used as an independent checker of the grid_sample statement and, being
OpenMP-parallel in the forward, as a fast CPU checker at large sizes.
"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "msda_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libmsda_ref.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB + ".tmp", SRC, "-lm"],
                   check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        vp, ci = ctypes.c_void_p, ctypes.c_int
        _lib.msda_ref_forward_f32.argtypes = [vp] * 5 + [ci] * 7 + [vp]
        _lib.msda_ref_forward_f32.restype = ci
        _lib.msda_ref_backward_f32.argtypes = [vp] * 6 + [ci] * 7 + [vp] * 4
        _lib.msda_ref_backward_f32.restype = ci
    return _lib


def _prep(value, shapes, start, loc, attn):
    value = value.detach().float().contiguous().cpu()
    shapes = shapes.detach().to(torch.int64).contiguous().cpu()
    start = start.detach().to(torch.int64).contiguous().cpu()
    loc = loc.detach().float().contiguous().cpu()
    attn = attn.detach().float().contiguous().cpu()
    N, S, M, D = value.shape
    _, Q, _, L, P, _ = loc.shape
    return value, shapes, start, loc, attn, (N, S, M, D, L, Q, P)


def forward(value, shapes, start, loc, attn):
    value, shapes, start, loc, attn, dims = _prep(value, shapes, start, loc, attn)
    N, S, M, D, L, Q, P = dims
    out = torch.empty(N, Q, M * D)
    rc = _load().msda_ref_forward_f32(value.data_ptr(), shapes.data_ptr(), start.data_ptr(),
                                      loc.data_ptr(), attn.data_ptr(), *dims, out.data_ptr())
    assert rc == 0
    return out


def backward(value, shapes, start, loc, attn, grad_out):
    value, shapes, start, loc, attn, dims = _prep(value, shapes, start, loc, attn)
    grad_out = grad_out.detach().float().contiguous().cpu()
    gv = torch.zeros_like(value)
    gl = torch.zeros_like(loc)
    ga = torch.zeros_like(attn)
    scratch = torch.zeros(value.numel(), dtype=torch.float64)
    rc = _load().msda_ref_backward_f32(value.data_ptr(), shapes.data_ptr(), start.data_ptr(),
                                       loc.data_ptr(), attn.data_ptr(), grad_out.data_ptr(),
                                       *dims, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                                       scratch.data_ptr())
    assert rc == 0
    return gv, gl, ga
