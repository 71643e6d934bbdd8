"""The encoder caller's prologue kernels: BEV rotation, camera-feature flattening."""
import ctypes
import math
import os

import torch
from torch.autograd.function import Function, once_differentiable

from .. import _lib
from ..ext import _ptr, _req
from ..functions import MultiScaleDeformableAttnFunction_fp32

from .. import modes as _modes



# ---------------------------------------------------------------------------
# The encoder's caller (PerceptionTransformer.get_bev_features): csrc/prologue.h
# ---------------------------------------------------------------------------

def rotation_theta(angle_deg, center, h, w):
    """The normalised 2 x 3 inverse affine matrix torchvision's ``rotate(img, angle,
    center=center)`` hands to ``grid_sample`` (``rotate`` negates the angle and re-centres
    ``center`` on the image centre; ``_gen_affine_grid`` divides row 0 by w/2 and row 1 by
    h/2), as 6 fp32 values computed with the same fp32 roundings."""
    import math
    cx, cy = 1.0 * (center[0] - w * 0.5), 1.0 * (center[1] - h * 0.5)
    rot = math.radians(-angle_deg)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx) + m[1] * (-cy)
    m[5] += m[3] * (-cx) + m[4] * (-cy)
    m[2] += cx
    m[5] += cy
    theta = torch.tensor(m, dtype=torch.float32).reshape(2, 3)
    return (theta / torch.tensor([[0.5 * w], [0.5 * h]], dtype=torch.float32)).reshape(-1).tolist()


def rotation_theta_device(angles_deg, center, h, w):
    """``rotation_theta`` for a DEVICE tensor of angles (bs,) -> (bs, 6) fp32 on the device: the same
    formulas in float64 device arithmetic, the same fp32 roundings at the end; no host synchronisation
    (a captured step follows the pose of every replayed frame)."""
    a = angles_deg.to(torch.float64)
    cx, cy = 1.0 * (center[0] - w * 0.5), 1.0 * (center[1] - h * 0.5)
    rot = -a * (math.pi / 180.0)
    ca, sa = torch.cos(rot), torch.sin(rot)
    # (a, b, c, d) = (cos, -sin, sin, cos); m = [d, -b, 0, -c, a, 0]
    m0, m1, m3, m4 = ca, sa, -sa, ca
    m2 = m0 * (-cx) + m1 * (-cy) + cx
    m5 = m3 * (-cx) + m4 * (-cy) + cy
    theta = torch.stack([m0, m1, m2, m3, m4, m5], -1).to(torch.float32)
    key = ("rot_scale", h, w, str(theta.device))
    scale = _CONST_CACHE.get(key)
    if scale is None:
        scale = _CONST_CACHE[key] = torch.tensor([0.5 * w] * 3 + [0.5 * h] * 3, dtype=torch.float32, device=theta.device)
    return (theta / scale).contiguous()


def rotate_bev(prev_bev, angles_deg, center, bev_h, bev_w):
    """prev_bev (Q, bs, C) -> a new tensor whose batch entry i is rotated by ``angles_deg[i]``
    about ``center`` (nearest, zero fill): ``bevmsda_rotate_bev_f32`` (transformer.py:146-156).
    The argument is not written to (the reference overwrites it in place)."""
    _req(prev_bev.is_cuda and prev_bev.dim() == 3, "bevmsda: prev_bev must be a (Q, bs, C) GPU tensor")
    Q, bs, C = prev_bev.shape
    _req(Q == bev_h * bev_w, "bevmsda: prev_bev rows != bev_h * bev_w")
    if prev_bev.dtype != torch.float32 or (torch.is_grad_enabled() and prev_bev.requires_grad):
        # differentiable / half-precision history: the same nearest-neighbour map as a torch gather
        # (the kernel is fp32, forward only)
        cols = []
        for i in range(bs):
            th = torch.tensor(rotation_theta(float(angles_deg[i]), center, bev_h, bev_w),
                              dtype=torch.float32, device=prev_bev.device).view(1, 2, 3)
            grid = torch.nn.functional.affine_grid(th, (1, 1, bev_h, bev_w), align_corners=False)
            img = prev_bev[:, i].reshape(bev_h, bev_w, C).permute(2, 0, 1)[None]
            rot = torch.nn.functional.grid_sample(img.float(), grid, mode="nearest", padding_mode="zeros",
                                                  align_corners=False)
            cols.append(rot[0].permute(1, 2, 0).reshape(Q, C).to(prev_bev.dtype))
        return torch.stack(cols, 1)
    src = prev_bev.contiguous()
    out = torch.empty_like(src)
    lib = _lib.load()
    if torch.is_tensor(angles_deg) and angles_deg.is_cuda:
        theta = rotation_theta_device(angles_deg.reshape(-1), center, bev_h, bev_w)
        with torch.cuda.device(src.device):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(bs):
                rc = lib.bevmsda_rotate_bev_dev_f32(src.data_ptr() + i * C * 4, bs * C, out.data_ptr() + i * C * 4,
                                                    bs * C, bev_h, bev_w, C, theta.data_ptr() + i * 24, st)
                _lib.check(rc, "rotate_bev (device pose)")
        return out
    with torch.cuda.device(src.device):
        st = torch.cuda.current_stream().cuda_stream
        for i in range(bs):
            theta = (ctypes.c_float * 6)(*rotation_theta(float(angles_deg[i]), center, bev_h, bev_w))
            rc = lib.bevmsda_rotate_bev_f32(src.data_ptr() + i * C * 4, bs * C, out.data_ptr() + i * C * 4,
                                            bs * C, bev_h, bev_w, C, theta, st)
            _lib.check(rc, "rotate_bev")
    return out


_CONST_CACHE = {}


def _level_tensors(shapes, device):
    """(spatial_shapes, level_start_index) int64 device tensors of a level shape list, built once per
    (shapes, device): a host -> device copy per frame is a latency bubble and cannot be captured in a HIP graph."""
    key = ("levels", shapes, str(device))
    hit = _CONST_CACHE.get(key)
    if hit is None:
        ss = torch.as_tensor(shapes, dtype=torch.long, device=device)
        hit = _CONST_CACHE[key] = (ss, torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1])))
    return hit


def _flatten_feats_torch(mlvl_feats, cams_embeds, level_embeds):
    """transformer.py:165-184 in torch ops (autograd / any dtype)."""
    flat, shapes = [], []
    for lvl, feat in enumerate(mlvl_feats):
        h, w = feat.shape[3:]
        feat = feat.flatten(3).permute(1, 0, 3, 2)
        if cams_embeds is not None:
            feat = feat + cams_embeds[:, None, None, :].to(feat.dtype)
        feat = feat + level_embeds[None, None, lvl:lvl + 1, :].to(feat.dtype)
        shapes.append((int(h), int(w)))
        flat.append(feat)
    out = torch.cat(flat, 2).permute(0, 2, 1, 3)
    spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=out.device)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    return out, spatial_shapes, level_start_index


def flatten_feats(mlvl_feats, cams_embeds, level_embeds):
    """list of (bs, Nc, C, h, w) -> feat_flatten (Nc, S, bs, C) with ``+ cams_embeds[cam]``
    (or None) ``+ level_embeds[lvl]``, plus spatial_shapes (L, 2) and level_start_index (L,)
    int64 device tensors (transformer.py:165-184): ``bevmsda_flatten_feats_f32``."""
    f0 = mlvl_feats[0]
    _req(f0.is_cuda, "bevmsda: camera features must be GPU tensors (there is no CPU path)")
    needs_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (*mlvl_feats, cams_embeds, level_embeds))
    if needs_grad or any(f.dtype != torch.float32 for f in mlvl_feats) or torch.is_autocast_enabled():
        # training (gradients flow to the backbone features and both embeddings) or mixed precision:
        # the reference's differentiable torch statements (transformer.py:165-184); the kernel below
        # writes into a fresh buffer and has no autograd graph
        return _flatten_feats_torch(mlvl_feats, cams_embeds, level_embeds)
    bs, Nc, C = f0.shape[:3]
    shapes = [(int(f.shape[3]), int(f.shape[4])) for f in mlvl_feats]
    S = sum(h * w for h, w in shapes)
    out = torch.empty((Nc, S, bs, C), dtype=torch.float32, device=f0.device)
    lib = _lib.load()
    ce = cams_embeds.float().contiguous() if cams_embeds is not None else None
    le = level_embeds.float().contiguous()
    s0 = 0
    with torch.cuda.device(f0.device):
        st = torch.cuda.current_stream().cuda_stream
        for lvl, (f, (h, w)) in enumerate(zip(mlvl_feats, shapes)):
            _req(tuple(f.shape[:3]) == (bs, Nc, C) and f.dtype == torch.float32,
                 "bevmsda: inconsistent feature levels")
            f = f.contiguous()
            rc = lib.bevmsda_flatten_feats_f32(_ptr(f), _ptr(ce) if ce is not None else None,
                                               le.data_ptr() + lvl * C * 4, _ptr(out), bs, Nc, C, h * w,
                                               S, s0, st)
            _lib.check(rc, "flatten_feats")
            s0 += h * w
    spatial_shapes, level_start_index = _level_tensors(tuple(map(tuple, shapes)), f0.device)
    return out, spatial_shapes, level_start_index
