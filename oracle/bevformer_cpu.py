"""TEST INFRASTRUCTURE — CPU restatement ("port") of the reference's BEV-encoder
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product path never does.

It restates, as plain functions over a reference-keyed ``state_dict`` and fp32
CPU tensors, what these reference files compute (all paths relative to
/root/reference/projects/mmdet3d_plugin/bevformer/modules/):

  encoder.py:46-85            reference points (3d pillar anchors / 2d BEV grid)
  encoder.py:88-149           camera projection + visibility mask
  encoder.py:151-239          encoder driver (hybrid prev/cur value, layer loop)
  encoder.py:287-406          layer op-order interpreter
  temporal_self_attention.py:128-272
  spatial_cross_attention.py:76-175, 273-399
  custom_base_transformer_layer.py:72-163   (FFN/LayerNorm construction)
  transformer.py:104-200      PerceptionTransformer.get_bev_features (the encoder's caller:
                              ego-motion shift, prev-BEV rotation, can-bus MLP, camera /
                              level embeddings, flatten) — SURVEY.md §8f rank 1
  decoder.py:53-129, 133-345  DetectionTransformerDecoder layer loop / reference-point
                              refinement, CustomMSDeformableAttention — SURVEY.md §8f rank 3
  mmcv-full 1.4.0 (not on disk): multi_scale_deformable_attn_pytorch, FFN
  torchvision 0.10.1 (not on disk; docs/install.md:15 pins torch 1.9.1 whose matching
                              torchvision is 0.10.1): transforms.functional.rotate — PARITY
                              UNPINNED for this one function: restated from its published
                              algorithm in ``rotate_nearest`` below; the reference's own
                              transformer.py is executed with this restatement bound to the
                              name ``rotate`` (oracle/mmcv_stub.py), which pins everything
                              around it but not its inside.

Pinning status: the reference ships no tests or golden vectors for this path
(SURVEY.md §4, §8c), so this restatement is pinned against the reference's own
files executed here under ``oracle/mmcv_stub.py`` (tests/test_oracle_vs_reference.py,
fixtures in tests/golden/ made by oracle/make_golden.py).  The third-party
deformable-attention CPU fallback is restated from its published algorithm
(Deformable-DETR: per-level ``grid_sample(bilinear, zeros, align_corners=False)``
on ``2*loc-1`` followed by the attention-weighted sum) and cross-checked
against an independent loop implementation (``msda_loops``, oracle/msda_ref.c)
and HuggingFace's copy of the same algorithm.
"""
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# multi-scale deformable attention (operator level)
# --------------------------------------------------------------------------

def msda_gridsample(value, value_spatial_shapes, sampling_locations, attention_weights):
    """mmcv's CPU fallback, called at spatial_cross_attention.py:394-395 and
    temporal_self_attention.py:252-253.

    value (N,S,M,D); value_spatial_shapes (L,2) as (H,W); sampling_locations
    (N,Q,M,L,P,2) as (x,y) in [0,1]; attention_weights (N,Q,M,L,P) -> (N,Q,M*D)."""
    N, _, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    grids = 2 * sampling_locations - 1
    per_level = []
    for lvl, chunk in enumerate(value.split(sizes, dim=1)):
        h, w = int(value_spatial_shapes[lvl][0]), int(value_spatial_shapes[lvl][1])
        fmap = chunk.flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        grid = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)          # (N*M,Q,P,2)
        per_level.append(F.grid_sample(fmap, grid, mode="bilinear",
                                       padding_mode="zeros", align_corners=False))
    sampled = torch.stack(per_level, dim=-2).flatten(-2)                   # (N*M,D,Q,L*P)
    w = attention_weights.transpose(1, 2).reshape(N * M, 1, Q, L * P)
    out = (sampled * w).sum(-1).view(N, M * D, Q)
    return out.transpose(1, 2).contiguous()


def msda_loops(value, value_spatial_shapes, level_start_index, sampling_locations,
               attention_weights):
    """Independent scalar-loop statement of the operator from its math
    (SURVEY.md Appendix A), in float64 numpy; small cases only.
    Also returns nothing else: gradients come from autograd of msda_gridsample."""
    v = value.detach().double().numpy()
    loc = sampling_locations.detach().double().numpy()
    att = attention_weights.detach().double().numpy()
    N, _, M, D = v.shape
    _, Q, _, L, P, _ = loc.shape
    out = np.zeros((N, Q, M, D))
    for n in range(N):
        for q in range(Q):
            for m in range(M):
                for l in range(L):
                    H, W = int(value_spatial_shapes[l][0]), int(value_spatial_shapes[l][1])
                    base = int(level_start_index[l])
                    for p in range(P):
                        x = loc[n, q, m, l, p, 0] * W - 0.5
                        y = loc[n, q, m, l, p, 1] * H - 0.5
                        x0, y0 = int(np.floor(x)), int(np.floor(y))
                        for (yy, xx) in ((y0, x0), (y0, x0 + 1), (y0 + 1, x0), (y0 + 1, x0 + 1)):
                            if 0 <= yy < H and 0 <= xx < W:
                                wgt = (1 - abs(y - yy)) * (1 - abs(x - xx))
                                out[n, q, m] += att[n, q, m, l, p] * wgt * v[n, base + yy * W + xx, m]
    return torch.from_numpy(out.reshape(N, Q, M * D))


def msda_backward_autograd(value, value_spatial_shapes, sampling_locations,
                           attention_weights, grad_output):
    """Gradients of the operator w.r.t. (value, sampling_locations,
    attention_weights) = what ``ms_deform_attn_backward`` accumulates
    (multi_scale_deformable_attn_function.py:146-163), via autograd of the
    CPU fallback."""
    v = value.detach().clone().requires_grad_(True)
    l = sampling_locations.detach().clone().requires_grad_(True)
    a = attention_weights.detach().clone().requires_grad_(True)
    out = msda_gridsample(v, value_spatial_shapes, l, a)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), grad_output)
    return gv, gl, ga


# --------------------------------------------------------------------------
# geometry (once per frame)
# --------------------------------------------------------------------------

def pillar_points(H, W, Z, num_z, bs, dtype=torch.float32):
    """encoder.py:61-71 -> (bs, num_z, H*W, 3) normalised (x, y, z)."""
    zs = torch.linspace(0.5, Z - 0.5, num_z, dtype=dtype).view(-1, 1, 1).expand(num_z, H, W) / Z
    xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype).view(1, 1, W).expand(num_z, H, W) / W
    ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype).view(1, H, 1).expand(num_z, H, W) / H
    pts = torch.stack((xs, ys, zs), -1).permute(0, 3, 1, 2).flatten(2).permute(0, 2, 1)
    return pts[None].repeat(bs, 1, 1, 1)


def bev_grid_points(H, W, bs, dtype=torch.float32):
    """encoder.py:74-85 -> (bs, H*W, 1, 2) normalised (x, y)."""
    ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype),
                            torch.linspace(0.5, W - 0.5, W, dtype=dtype), indexing="ij")
    ref = torch.stack((rx.reshape(-1)[None] / W, ry.reshape(-1)[None] / H), -1)
    return ref.repeat(bs, 1, 1).unsqueeze(2)


def project_to_cameras(ref_3d, pc_range, img_metas):
    """encoder.py:95-144 -> reference_points_cam (Nc,B,Q,Dz,2), bev_mask (Nc,B,Q,Dz)."""
    l2i = ref_3d.new_tensor(np.asarray([m["lidar2img"] for m in img_metas]))  # (B,Nc,4,4)
    p = ref_3d.clone()
    p[..., 0:1] = p[..., 0:1] * (pc_range[3] - pc_range[0]) + pc_range[0]
    p[..., 1:2] = p[..., 1:2] * (pc_range[4] - pc_range[1]) + pc_range[1]
    p[..., 2:3] = p[..., 2:3] * (pc_range[5] - pc_range[2]) + pc_range[2]
    p = torch.cat((p, torch.ones_like(p[..., :1])), -1).permute(1, 0, 2, 3)   # (Dz,B,Q,4)
    Dz, B, Q = p.shape[:3]
    Nc = l2i.size(1)
    p = p.view(Dz, B, 1, Q, 4).repeat(1, 1, Nc, 1, 1).unsqueeze(-1)
    mats = l2i.view(1, B, Nc, 1, 4, 4).repeat(Dz, 1, 1, Q, 1, 1)
    cam = torch.matmul(mats.float(), p.float()).squeeze(-1)
    eps = 1e-5
    mask = cam[..., 2:3] > eps
    cam = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
    cam[..., 0] /= img_metas[0]["img_shape"][0][1]
    cam[..., 1] /= img_metas[0]["img_shape"][0][0]
    mask = (mask & (cam[..., 1:2] > 0.0) & (cam[..., 1:2] < 1.0)
            & (cam[..., 0:1] < 1.0) & (cam[..., 0:1] > 0.0))
    mask = torch.nan_to_num(mask)
    return cam.permute(2, 1, 3, 0, 4), mask.permute(2, 1, 3, 0, 4).squeeze(-1)


# --------------------------------------------------------------------------
# modules as functions over a state_dict
# --------------------------------------------------------------------------

def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _drop(drop, x):
    """An nn.Dropout site of the reference in train() mode: ``drop`` yields the scale tensor (0 or 1 / (1 - p) per element)
    of each site in the order the reference executes them; None = eval mode (identity)."""
    return x if drop is None else x * next(drop).to(x.dtype).reshape(x.shape)


def temporal_self_attention(sd, pre, query, value, bev_pos, ref_2d, bev_h, bev_w,
                            num_heads=8, num_points=4, msda=msda_gridsample, drop=None):
    """temporal_self_attention.py:177-272 (``drop``: the scale tensors of ``self.dropout`` at :272 in train() mode)."""
    bs, Q, C = query.shape
    if value is None:
        value = torch.stack([query, query], 1).reshape(bs * 2, Q, C)
    identity = query
    if bev_pos is not None:
        query = query + bev_pos
    shapes = torch.tensor([[bev_h, bev_w]])
    q2 = torch.cat([value[:bs], query], -1)
    v = _lin(sd, pre + "value_proj", value).reshape(bs * 2, value.shape[1], num_heads, -1)
    off = _lin(sd, pre + "sampling_offsets", q2).view(bs, Q, num_heads, 2, 1, num_points, 2)
    att = _lin(sd, pre + "attention_weights", q2).view(bs, Q, num_heads, 2, num_points)
    att = att.softmax(-1).view(bs, Q, num_heads, 2, 1, num_points)
    att = att.permute(0, 3, 1, 2, 4, 5).reshape(bs * 2, Q, num_heads, 1, num_points).contiguous()
    off = off.permute(0, 3, 1, 2, 4, 5, 6).reshape(bs * 2, Q, num_heads, 1, num_points, 2)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref_2d[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = msda(v, shapes, loc, att)                                     # (bs*2,Q,C)
    out = out.permute(1, 2, 0).view(Q, C, bs, 2).mean(-1).permute(2, 0, 1)
    return _drop(drop, _lin(sd, pre + "output_proj", out)) + identity


def deformable_attention_3d(sd, pre, query, value, ref_cam, shapes, num_heads=8,
                            num_points=8, msda=msda_gridsample):
    """spatial_cross_attention.py:318-399 with batch_first=True."""
    bs, Q, _ = query.shape
    L = shapes.shape[0]
    v = _lin(sd, pre + "value_proj", value).view(bs, value.shape[1], num_heads, -1)
    off = _lin(sd, pre + "sampling_offsets", query).view(bs, Q, num_heads, L, num_points, 2)
    att = _lin(sd, pre + "attention_weights", query).view(bs, Q, num_heads, L * num_points)
    att = att.softmax(-1).view(bs, Q, num_heads, L, num_points)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    Dz = ref_cam.shape[2]
    off = off / norm[None, None, None, :, None, :]
    off = off.view(bs, Q, num_heads, L, num_points // Dz, Dz, 2)
    loc = (ref_cam[:, :, None, None, None, :, :] + off).view(bs, Q, num_heads, L, num_points, 2)
    return msda(v, shapes, loc, att)


def spatial_cross_attention(sd, pre, query, feats, ref_cam, bev_mask, shapes,
                            msda=msda_gridsample, drop=None):
    """spatial_cross_attention.py:123-175 (``drop``: ``self.dropout`` at :175 in train() mode), padded per-camera
    rebatch exactly as the reference does it, visibility taken from batch element 0."""
    bs, Q, C = query.shape
    Nc = feats.shape[0]
    Dz = ref_cam.size(3)
    residual = query
    slots = torch.zeros_like(query)
    idx = [m[0].sum(-1).nonzero().squeeze(-1) for m in bev_mask]
    max_len = max(len(i) for i in idx)
    q_re = query.new_zeros(bs, Nc, max_len, C)
    r_re = ref_cam.new_zeros(bs, Nc, max_len, Dz, 2)
    for j in range(bs):
        for i in range(Nc):
            q_re[j, i, :len(idx[i])] = query[j, idx[i]]
            r_re[j, i, :len(idx[i])] = ref_cam[i][j, idx[i]]
    S = feats.shape[1]
    val = feats.permute(2, 0, 1, 3).reshape(bs * Nc, S, C)
    out = deformable_attention_3d(sd, pre + "deformable_attention.", q_re.view(bs * Nc, max_len, C),
                                  val, r_re.view(bs * Nc, max_len, Dz, 2), shapes, msda=msda)
    out = out.view(bs, Nc, max_len, C)
    for j in range(bs):
        for i in range(Nc):
            slots[j, idx[i]] += out[j, i, :len(idx[i])]
    count = (bev_mask.sum(-1) > 0).permute(1, 2, 0).sum(-1)
    slots = slots / torch.clamp(count, min=1.0)[..., None]
    return _drop(drop, _lin(sd, pre + "output_proj", slots)) + residual


def ffn(sd, pre, x, drop=None):
    """mmcv FFN (add_identity=True), keys layers.0.0 / layers.1: Linear, ReLU, Dropout, Linear, Dropout."""
    h = _drop(drop, F.relu(_lin(sd, pre + "layers.0.0", x)))
    return x + _drop(drop, _lin(sd, pre + "layers.1", h))


def layer_norm(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def encoder_layer(sd, pre, query, feats, bev_pos, ref_2d, ref_cam, bev_mask, bev_h, bev_w,
                  shapes, prev_bev, msda=msda_gridsample, drop=None):
    """encoder.py:356-406 with operation_order
    ('self_attn','norm','cross_attn','norm','ffn','norm')."""
    x = temporal_self_attention(sd, pre + "attentions.0.", query, prev_bev, bev_pos, ref_2d,
                                bev_h, bev_w, msda=msda, drop=drop)
    x = layer_norm(sd, pre + "norms.0", x)
    x = spatial_cross_attention(sd, pre + "attentions.1.", x, feats, ref_cam, bev_mask, shapes,
                                msda=msda, drop=drop)
    x = layer_norm(sd, pre + "norms.1", x)
    x = ffn(sd, pre + "ffns.0.", x, drop=drop)
    return layer_norm(sd, pre + "norms.2", x)


def encoder_forward(sd, bev_query, feats, *, bev_h, bev_w, bev_pos, spatial_shapes,
                    level_start_index=None, prev_bev=None, shift=None, img_metas=None,
                    num_layers=None, pc_range=None, num_points_in_pillar=4,
                    msda=msda_gridsample, return_intermediate=False, dropout_scales=None, **_):
    """encoder.py:185-239.  ``bev_query``/``bev_pos``/``prev_bev`` are (Q,bs,C),
    ``feats`` is (Nc,S,bs,C); returns (bs,Q,C).  ``dropout_scales``: train() mode — the scale tensors of the four
    nn.Dropout sites of every layer in execution order (TSA output, SCA output, FFN hidden, FFN output)."""
    drop = iter(dropout_scales) if dropout_scales is not None else None
    if num_layers is None:
        num_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    bs = bev_query.size(1)
    ref_3d = pillar_points(bev_h, bev_w, pc_range[5] - pc_range[2], num_points_in_pillar, bs,
                           bev_query.dtype)
    ref_2d = bev_grid_points(bev_h, bev_w, bs, bev_query.dtype)
    ref_cam, bev_mask = project_to_cameras(ref_3d, pc_range, img_metas)
    shifted = ref_2d.clone() + shift[:, None, None, :]
    x = bev_query.permute(1, 0, 2)
    pos = bev_pos.permute(1, 0, 2)
    Q = x.shape[1]
    if prev_bev is not None:
        prev = torch.stack([prev_bev.permute(1, 0, 2), x], 1).reshape(bs * 2, Q, -1)
        hybrid = torch.stack([shifted, ref_2d], 1).reshape(bs * 2, Q, 1, 2)
    else:
        prev = None
        hybrid = torch.stack([ref_2d, ref_2d], 1).reshape(bs * 2, Q, 1, 2)
    inter = []
    for i in range(num_layers):
        x = encoder_layer(sd, f"layers.{i}.", x, feats, pos, hybrid, ref_cam, bev_mask,
                          bev_h, bev_w, spatial_shapes, prev, msda=msda, drop=drop)
        inter.append(x)
    return torch.stack(inter) if return_intermediate else x


# --------------------------------------------------------------------------
# PerceptionTransformer.get_bev_features (transformer.py:104-200)
# --------------------------------------------------------------------------

def _inverse_affine_matrix(center, angle_deg):
    """torchvision.transforms.functional._get_inverse_affine_matrix for a pure rotation
    (translate 0, scale 1, shear 0): [d, -b, ., -c, a, .] with the centre terms."""
    rot = np.deg2rad(angle_deg)
    cx, cy = center
    a, b, c, d = np.cos(rot), -np.sin(rot), np.sin(rot), np.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx) + m[1] * (-cy)
    m[5] += m[3] * (-cx) + m[4] * (-cy)
    m[2] += cx
    m[5] += cy
    return [float(v) for v in m]


def rotate_affine_grid(h, w, angle_deg, center, device="cpu"):
    """The normalised sampling grid (h*w, 2) torchvision 0.10's ``rotate(img (C,h,w), angle, center=center)`` hands
    to ``grid_sample``: ``rotate`` negates the angle and re-centres ``center`` on the image centre,
    ``_get_inverse_affine_matrix`` gives the 2 x 3 matrix, ``_gen_affine_grid`` builds the output grid from pixel
    centres (fp32 linspace, bmm with the matrix divided by (w/2, h/2))."""
    center_f = [1.0 * (c - s * 0.5) for c, s in zip(center, [w, h])]
    theta = torch.tensor(_inverse_affine_matrix(center_f, -angle_deg), dtype=torch.float32,
                         device=device).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=torch.float32, device=device)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w, device=device))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h,
                                      device=device).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32,
                                                    device=device)
    return base.view(1, h * w, 3).bmm(rescaled).view(h * w, 2)


def rotate_source_index(h, w, angle_deg, center, device="cpu"):
    """Source pixel of every output pixel of ``rotate(img (C,h,w), angle, center=center)``
    with nearest interpolation, as flat indices (h*w,) int64, -1 where the source falls
    outside the image (zero fill): ``grid_sample(nearest, zeros, align_corners=False)`` on
    ``rotate_affine_grid`` un-normalises with ((g + 1) * size - 1) / 2 and rounds half to even
    (pinned against torch's own grid_sample kernel in tests/test_transformer_cpu.py)."""
    grid = rotate_affine_grid(h, w, angle_deg, center, device)
    ix = torch.round(((grid[:, 0] + 1) * w - 1) / 2)          # torch.round: half to even
    iy = torch.round(((grid[:, 1] + 1) * h - 1) / 2)
    ok = (ix >= 0) & (ix <= w - 1) & (iy >= 0) & (iy <= h - 1)
    idx = (iy * w + ix).long()
    return torch.where(ok, idx, torch.full_like(idx, -1))


def rotate_nearest(img, angle_deg, center):
    """``torchvision.transforms.functional.rotate(img (C,h,w), angle, center=center)`` with
    its defaults (nearest, no expand, zero fill) — transformer.py:152-153."""
    C, h, w = img.shape
    idx = rotate_source_index(h, w, angle_deg, center, img.device)
    flat = img.reshape(C, h * w)
    out = flat[:, idx.clamp(min=0)] * (idx >= 0).to(img.dtype)
    return out.view(C, h, w)


def bev_shift(img_metas, bev_h, bev_w, grid_length=(0.512, 0.512), use_shift=True):
    """Ego-motion shift of the BEV reference points (transformer.py:123-141) -> (bs, 2)
    float64 numpy array (x, y)."""
    delta_x = np.array([m["can_bus"][0] for m in img_metas])
    delta_y = np.array([m["can_bus"][1] for m in img_metas])
    ego_angle = np.array([m["can_bus"][-2] / np.pi * 180 for m in img_metas])
    translation_length = np.sqrt(delta_x ** 2 + delta_y ** 2)
    translation_angle = np.arctan2(delta_y, delta_x) / np.pi * 180
    bev_angle = ego_angle - translation_angle
    shift_y = translation_length * np.cos(bev_angle / 180 * np.pi) / grid_length[0] / bev_h
    shift_x = translation_length * np.sin(bev_angle / 180 * np.pi) / grid_length[1] / bev_w
    return np.stack([shift_x * use_shift, shift_y * use_shift], -1)


def get_bev_features(tsd, esd, mlvl_feats, bev_queries, bev_h, bev_w, *, bev_pos, img_metas,
                     pc_range, grid_length=(0.512, 0.512), prev_bev=None, rotate_prev_bev=True,
                     use_shift=True, use_can_bus=True, can_bus_norm=True, use_cams_embeds=True,
                     rotate_center=(100, 100), rotate_fn=rotate_nearest):
    """PerceptionTransformer.get_bev_features (transformer.py:104-200).

    ``tsd``: the transformer's own parameters (``level_embeds``, ``cams_embeds``,
    ``can_bus_mlp.{0,2,norm}.*``); ``esd``: the encoder's state_dict (keys without the
    ``encoder.`` prefix).  mlvl_feats: list of (bs, Nc, C, h, w); bev_queries (Q, C);
    bev_pos (bs, C, bev_h, bev_w); prev_bev (Q, bs, C) or (bs, Q, C) or None -> (bs, Q, C)."""
    bs = mlvl_feats[0].size(0)
    bev_queries = bev_queries.unsqueeze(1).repeat(1, bs, 1)
    bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
    shift = bev_queries.new_tensor(bev_shift(img_metas, bev_h, bev_w, grid_length, use_shift))
    if prev_bev is not None:
        if prev_bev.shape[1] == bev_h * bev_w:
            prev_bev = prev_bev.permute(1, 0, 2)
        if rotate_prev_bev:
            prev_bev = prev_bev.clone()
            for i in range(bs):
                angle = img_metas[i]["can_bus"][-1]
                tmp = prev_bev[:, i].reshape(bev_h, bev_w, -1).permute(2, 0, 1)
                tmp = rotate_fn(tmp, angle, center=list(rotate_center))
                prev_bev[:, i] = tmp.permute(1, 2, 0).reshape(bev_h * bev_w, -1)
    can_bus = bev_queries.new_tensor(np.array([m["can_bus"] for m in img_metas]))
    x = F.relu(F.linear(can_bus, tsd["can_bus_mlp.0.weight"], tsd["can_bus_mlp.0.bias"]))
    x = F.relu(F.linear(x, tsd["can_bus_mlp.2.weight"], tsd["can_bus_mlp.2.bias"]))
    if can_bus_norm:
        x = F.layer_norm(x, (x.shape[-1],), tsd["can_bus_mlp.norm.weight"],
                         tsd["can_bus_mlp.norm.bias"])
    bev_queries = bev_queries + x[None] * use_can_bus
    flat, shapes = [], []
    for lvl, feat in enumerate(mlvl_feats):
        _, num_cam, c, h, w = feat.shape
        feat = feat.flatten(3).permute(1, 0, 3, 2)
        if use_cams_embeds:
            feat = feat + tsd["cams_embeds"][:, None, None, :].to(feat.dtype)
        feat = feat + tsd["level_embeds"][None, None, lvl:lvl + 1, :].to(feat.dtype)
        shapes.append((h, w))
        flat.append(feat)
    feat_flatten = torch.cat(flat, 2).permute(0, 2, 1, 3)
    spatial_shapes = torch.as_tensor(shapes, dtype=torch.long)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)),
                                   spatial_shapes.prod(1).cumsum(0)[:-1]))
    return encoder_forward(esd, bev_queries, feat_flatten, bev_h=bev_h, bev_w=bev_w,
                           bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                           level_start_index=level_start_index, prev_bev=prev_bev, shift=shift,
                           img_metas=img_metas, pc_range=pc_range)


# --------------------------------------------------------------------------
# decoder.py: CustomMSDeformableAttention (:133-345), DetectionTransformerDecoder (:53-129)
# --------------------------------------------------------------------------

def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def custom_ms_deformable_attention(sd, pre, query, value, reference_points, spatial_shapes, *,
                                   query_pos=None, identity=None, num_heads=8, num_levels=1,
                                   num_points=4, batch_first=False, msda=msda_gridsample):
    """decoder.py:247-345 (eval mode: dropout is the identity).  query (nq, bs, C), value
    (nv, bs, C) unless batch_first; reference_points (bs, nq, L, 2 | 4)."""
    if value is None:
        value = query
    if identity is None:
        identity = query
    if query_pos is not None:
        query = query + query_pos
    if not batch_first:
        query = query.permute(1, 0, 2)
        value = value.permute(1, 0, 2)
    bs, nq, _ = query.shape
    nv = value.shape[1]
    M, L, P = num_heads, num_levels, num_points
    v = _lin(sd, pre + "value_proj", value).view(bs, nv, M, -1)
    off = _lin(sd, pre + "sampling_offsets", query).view(bs, nq, M, L, P, 2)
    att = _lin(sd, pre + "attention_weights", query).view(bs, nq, M, L * P).softmax(-1)
    att = att.view(bs, nq, M, L, P)
    if reference_points.shape[-1] == 2:
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] \
            + off / P * reference_points[:, :, None, :, None, 2:] * 0.5
    out = _lin(sd, pre + "output_proj", msda(v, spatial_shapes, loc, att))
    if not batch_first:
        out = out.permute(1, 0, 2)
    return out + identity


def detection_decoder(layer_fns, query, reference_points, reg_branches=None,
                      return_intermediate=False):
    """decoder.py:89-129 over ``layer_fns[lid](output, reference_points_input)``."""
    output, inter, inter_ref = query, [], []
    for lid, fn in enumerate(layer_fns):
        output = fn(output, reference_points[..., :2].unsqueeze(2)).permute(1, 0, 2)
        if reg_branches is not None:
            tmp = reg_branches[lid](output)
            new = torch.zeros_like(reference_points)
            new[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points[..., :2])
            new[..., 2:3] = tmp[..., 4:5] + inverse_sigmoid(reference_points[..., 2:3])
            reference_points = new.sigmoid().detach()
        output = output.permute(1, 0, 2)
        if return_intermediate:
            inter.append(output)
            inter_ref.append(reference_points)
    if return_intermediate:
        return torch.stack(inter), torch.stack(inter_ref)
    return output, reference_points


# ---------------------------------------------------------------------------------------------
# History-BEV queue (detector level): detectors/bevformer.py:158-177 and :236-269
# ---------------------------------------------------------------------------------------------
def obtain_history_bev(bev_fn, feats_queue, img_metas_list):
    """``BEVFormer.obtain_history_bev`` (detectors/bevformer.py:158-177) with the head call
    ``self.pts_bbox_head(img_feats, img_metas, prev_bev, only_bev=True)`` (:175-176) abstracted as
    ``bev_fn``: frames in order, no gradients, scene reset on ``prev_bev_exists`` (:171-172)."""
    import torch
    with torch.no_grad():
        prev_bev = None
        len_queue = feats_queue[0].shape[1]
        for i in range(len_queue):
            img_metas = [each[i] for each in img_metas_list]
            if not img_metas[0]["prev_bev_exists"]:
                prev_bev = None
            img_feats = [each_scale[:, i] for each_scale in feats_queue]
            prev_bev = bev_fn(img_feats, img_metas, prev_bev)
        return prev_bev


def forward_test_step(prev_frame_info, bev_fn, mlvl_feats, img_metas, video_test_mode=True):
    """The state update of ``BEVFormer.forward_test`` (detectors/bevformer.py:236-269) around
    ``simple_test`` (abstracted as ``bev_fn``); like the reference it rewrites
    ``img_metas[0]['can_bus']`` IN PLACE and mutates ``prev_frame_info``.  Returns the new BEV."""
    import copy
    if img_metas[0]["scene_token"] != prev_frame_info["scene_token"]:
        prev_frame_info["prev_bev"] = None
    prev_frame_info["scene_token"] = img_metas[0]["scene_token"]
    if not video_test_mode:
        prev_frame_info["prev_bev"] = None
    tmp_pos = copy.deepcopy(img_metas[0]["can_bus"][:3])
    tmp_angle = copy.deepcopy(img_metas[0]["can_bus"][-1])
    if prev_frame_info["prev_bev"] is not None:
        img_metas[0]["can_bus"][:3] -= prev_frame_info["prev_pos"]
        img_metas[0]["can_bus"][-1] -= prev_frame_info["prev_angle"]
    else:
        img_metas[0]["can_bus"][-1] = 0
        img_metas[0]["can_bus"][:3] = 0
    new_prev_bev = bev_fn(mlvl_feats, img_metas, prev_frame_info["prev_bev"])
    prev_frame_info["prev_pos"] = tmp_pos
    prev_frame_info["prev_angle"] = tmp_angle
    prev_frame_info["prev_bev"] = new_prev_bev
    return new_prev_bev
