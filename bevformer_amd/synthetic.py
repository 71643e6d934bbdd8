"""Seeded synthetic workloads for the BEV-encoder path.

There is no nuScenes data and no checkpoint in this environment, so every
parity test, golden fixture and bench line is driven from the generators
below: a 6-pinhole camera rig that plays the role of ``img_metas['lidar2img']``
(reference contract: projects/mmdet3d_plugin/datasets/nuscenes_dataset.py:129-139),
N(0,1) camera features / BEV queries / BEV positional encodings, and two
weight regimes.  Everything is generated on CPU in fp32 from an explicit
``torch.Generator`` so that the same call reproduces the same bytes here and
on the GPU box (same image, same torch build).
"""
import math

import numpy as np
import torch

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]  # bevformer_base.py:9

# name -> geometry of one encoder call.  ``shapes`` are the (H, W) of the
# flattened camera feature levels, ``img`` the padded image (H, W), ``s`` the
# intrinsics scale (RandomScaleImageMultiViewImage), citations in SURVEY §8a-T1.
WORKLOADS = {
    # unit-test sized rig: small enough for the pure-python loops in tests
    "micro": dict(bev_h=12, bev_w=10, layers=2, shapes=[(8, 13)], img=(256, 416), s=0.26),
    "micro4": dict(bev_h=12, bev_w=10, layers=2, shapes=[(16, 26), (8, 13), (4, 7), (2, 4)],
                   img=(256, 416), s=0.26),
    # projects/configs/bevformer/bevformer_tiny.py:45-47,90,184-185
    "tiny": dict(bev_h=50, bev_w=50, layers=3, shapes=[(15, 25)], img=(480, 800), s=0.5),
    # projects/configs/bevformer/bevformer_small.py:41-43,88,182-183
    "small": dict(bev_h=150, bev_w=150, layers=3, shapes=[(23, 40)], img=(736, 1280), s=0.8),
    # BASELINE.json configs[2]: "150x150 BEV, 4 levels" (synthetic shape set)
    "small4": dict(bev_h=150, bev_w=150, layers=3,
                   shapes=[(92, 160), (46, 80), (23, 40), (12, 20)], img=(736, 1280), s=0.8),
    # projects/configs/bevformer/bevformer_base.py:34-37,54-61,80
    "base": dict(bev_h=200, bev_w=200, layers=6,
                 shapes=[(116, 200), (58, 100), (29, 50), (15, 25)], img=(928, 1600), s=1.0),
    # one layer of the base geometry: the gradient checks against autograd through the CPU oracle (six layers of
    # saved grid_sample intermediates do not fit a test's time budget, one does)
    "base1": dict(bev_h=200, bev_w=200, layers=1,
                  shapes=[(116, 200), (58, 100), (29, 50), (15, 25)], img=(928, 1600), s=1.0),
}

EMBED_DIMS = 256
NUM_CAMS = 6
_YAWS_DEG = (0.0, -55.0, 55.0, 180.0, -110.0, 110.0)


def encoder_cfg(name):
    """The ``encoder=dict(...)`` block of the reference configs
    (projects/configs/bevformer/bevformer_base.py:78-105) for workload ``name``."""
    w = WORKLOADS[name]
    return dict(
        type="BEVFormerEncoder",
        num_layers=w["layers"],
        pc_range=list(PC_RANGE),
        num_points_in_pillar=4,
        return_intermediate=False,
        transformerlayers=dict(
            type="BEVFormerLayer",
            attn_cfgs=[
                dict(type="TemporalSelfAttention", embed_dims=EMBED_DIMS, num_levels=1),
                dict(type="SpatialCrossAttention", pc_range=list(PC_RANGE),
                     deformable_attention=dict(type="MSDeformableAttention3D",
                                               embed_dims=EMBED_DIMS, num_points=8,
                                               num_levels=len(w["shapes"])),
                     embed_dims=EMBED_DIMS),
            ],
            feedforward_channels=EMBED_DIMS * 2,
            ffn_dropout=0.1,
            operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"),
        ),
    )


def camera_rig(scale):
    """Six pinhole cameras around the ego vehicle -> list of 4x4 float64
    ``lidar2img`` matrices (K @ E), as the dataset would hand them over."""
    mats = []
    for i, yaw in enumerate(_YAWS_DEG):
        psi = math.radians(yaw)
        fwd = np.array([math.cos(psi), math.sin(psi), 0.0])
        right = np.array([math.sin(psi), -math.cos(psi), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd])
        c = np.array([0.0, 0.0, -0.3])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ c
        f = (809.2 if i == 3 else 1266.4) * scale
        K = np.eye(4)
        K[0, 0] = f
        K[1, 1] = f
        K[0, 2] = 816.3 * scale
        K[1, 2] = 491.5 * scale
        mats.append(K @ E)
    return mats


def make_img_metas(name, bs=1):
    w = WORKLOADS[name]
    H, W = w["img"]
    return [dict(lidar2img=camera_rig(w["s"]), img_shape=[(H, W, 3)] * NUM_CAMS)
            for _ in range(bs)]


def level_tensors(name, device="cpu"):
    shapes = torch.tensor(WORKLOADS[name]["shapes"], dtype=torch.long, device=device)
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    return shapes, start


def make_inputs(name, seed=0, bs=1, temporal=False, device="cpu", dtype=torch.float32):
    """Keyword arguments of one ``BEVFormerEncoder.forward`` call
    (call site: projects/mmdet3d_plugin/bevformer/modules/transformer.py:186-198).

    Returns ``(bev_query, feat_flatten, kwargs)``; with ``temporal`` a random
    ``prev_bev`` and the ego-motion ``shift`` of SURVEY §8d are included."""
    w = WORKLOADS[name]
    g = torch.Generator().manual_seed(seed)
    Q = w["bev_h"] * w["bev_w"]
    S = sum(h * ww for h, ww in w["shapes"])
    bev_query = torch.randn(Q, bs, EMBED_DIMS, generator=g)
    bev_pos = torch.randn(Q, bs, EMBED_DIMS, generator=g)
    feat = torch.randn(NUM_CAMS, S, bs, EMBED_DIMS, generator=g)
    prev_bev = torch.randn(Q, bs, EMBED_DIMS, generator=g) if temporal else None
    shift = torch.zeros(bs, 2)
    if temporal:
        shift[:, 1] = 2.5 / 0.512 / w["bev_h"]
    shapes, start = level_tensors(name, device)

    def mv(t):
        return None if t is None else t.to(device=device, dtype=dtype)

    kwargs = dict(bev_h=w["bev_h"], bev_w=w["bev_w"], bev_pos=mv(bev_pos),
                  spatial_shapes=shapes, level_start_index=start,
                  prev_bev=mv(prev_bev), shift=mv(shift),
                  img_metas=make_img_metas(name, bs))
    return mv(bev_query), mv(feat), kwargs


def decoder_cfg(num_layers=2, num_levels=1):
    """A ``DetectionTransformerDecoder`` config in the style of bevformer_base.py:106-130 whose
    layers only use classes that live in the reference tree: cross-attention into the BEV grid
    (``CustomMSDeformableAttention``), norm, FFN, norm (the configs' self-attention is mmcv's
    ``MultiheadAttention`` inside mmdet's ``DetrTransformerDecoderLayer`` — third-party)."""
    return dict(
        type="DetectionTransformerDecoder", num_layers=num_layers, return_intermediate=True,
        transformerlayers=dict(
            type="MyCustomBaseTransformerLayer", batch_first=False,     # (num_query, bs, C) layout,
            attn_cfgs=[dict(type="CustomMSDeformableAttention", embed_dims=EMBED_DIMS,   # as the decoder
                            num_levels=num_levels)],
            feedforward_channels=EMBED_DIMS * 2, ffn_dropout=0.1,
            operation_order=("cross_attn", "norm", "ffn", "norm")))


def reference_decoder_cfg(num_layers=6):
    """The ``decoder=dict(...)`` block of projects/configs/bevformer/bevformer_base.py:106-127
    verbatim (third-party layer types: built by mmcv / mmdet when installed, by this package's
    restatements otherwise)."""
    return dict(
        type="DetectionTransformerDecoder", num_layers=num_layers, return_intermediate=True,
        transformerlayers=dict(
            type="DetrTransformerDecoderLayer",
            attn_cfgs=[dict(type="MultiheadAttention", embed_dims=EMBED_DIMS, num_heads=8, dropout=0.1),
                       dict(type="CustomMSDeformableAttention", embed_dims=EMBED_DIMS, num_levels=1)],
            feedforward_channels=EMBED_DIMS * 2, ffn_dropout=0.1,
            operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")))


def make_decoder_inputs(bev_h, bev_w, num_query=37, bs=2, seed=0, device="cpu"):
    """query / query_pos (nq, bs, C), value = bev_embed (Q, bs, C), reference_points
    (bs, nq, 3) in (0, 1), the (1, 2) / (1,) BEV level tensors (transformer.py:274-284)."""
    g = torch.Generator().manual_seed(seed)
    Q = bev_h * bev_w
    query = torch.randn(num_query, bs, EMBED_DIMS, generator=g)
    query_pos = torch.randn(num_query, bs, EMBED_DIMS, generator=g)
    value = torch.randn(Q, bs, EMBED_DIMS, generator=g)
    ref = torch.rand(bs, num_query, 3, generator=g) * 0.9 + 0.05
    shapes = torch.tensor([[bev_h, bev_w]], dtype=torch.long)
    start = torch.tensor([0], dtype=torch.long)
    return [t.to(device) for t in (query, query_pos, value, ref, shapes, start)]


def transformer_cfg(name):
    """The ``transformer=dict(type='PerceptionTransformer', ...)`` block of the reference
    configs (projects/configs/bevformer/bevformer_base.py:70-106) around ``encoder_cfg(name)``,
    without a decoder; ``rotate_center`` is the grid centre (the configs' [100, 100] for the
    200 x 200 base grid)."""
    w = WORKLOADS[name]
    return dict(type="PerceptionTransformer", num_feature_levels=len(w["shapes"]),
                num_cams=NUM_CAMS, rotate_prev_bev=True, use_shift=True, use_can_bus=True,
                embed_dims=EMBED_DIMS, rotate_center=[w["bev_w"] // 2, w["bev_h"] // 2],
                encoder=encoder_cfg(name))


def make_can_bus(seed=0, yaw_delta_deg=4.0):
    """An 18-float ``can_bus`` vector with the entries the path reads
    (datasets/nuscenes_dataset.py:148-165): [0:2] ego translation since the previous frame
    (m), [-2] ego yaw (rad), [-1] yaw delta (deg); the rest only feeds the can-bus MLP."""
    g = np.random.default_rng(seed)
    cb = g.standard_normal(18) * 0.1
    cb[0], cb[1] = 2.0, 1.5
    cb[-2] = 0.3
    cb[-1] = yaw_delta_deg
    return cb


def make_transformer_inputs(name, seed=0, bs=1, temporal=False, device="cpu"):
    """Arguments of one ``PerceptionTransformer.get_bev_features`` call (call site:
    dense_heads/bevformer_head.py:150-160): (mlvl_feats, bev_queries, kwargs)."""
    w = WORKLOADS[name]
    g = torch.Generator().manual_seed(seed)
    Q = w["bev_h"] * w["bev_w"]
    mlvl = [torch.randn(bs, NUM_CAMS, EMBED_DIMS, h, ww, generator=g) for h, ww in w["shapes"]]
    bev_queries = torch.randn(Q, EMBED_DIMS, generator=g)
    bev_pos = torch.randn(bs, EMBED_DIMS, w["bev_h"], w["bev_w"], generator=g)
    prev_bev = torch.randn(bs, Q, EMBED_DIMS, generator=g) if temporal else None
    metas = make_img_metas(name, bs)
    for i, m in enumerate(metas):
        m["can_bus"] = make_can_bus(seed + i, yaw_delta_deg=4.0 + 3.0 * i)

    def mv(t):
        return None if t is None else t.to(device)

    kwargs = dict(bev_h=w["bev_h"], bev_w=w["bev_w"], grid_length=(0.512, 0.512), bev_pos=mv(bev_pos),
                  prev_bev=mv(prev_bev), img_metas=metas)
    return [mv(f) for f in mlvl], mv(bev_queries), kwargs


def trained_like_(state_dict, seed=1):
    """Weight regime (ii) of SURVEY §8d, applied in place to a reference-keyed
    ``state_dict``: the reference initialisation is degenerate for parity work
    (zero offset/attention weights => data-independent sampling, uniform
    attention), so every Linear gets xavier-uniform weights, the sampling-offset
    and attention-weight projections get small random weights (bias of the
    offsets keeps the head-direction grid) and biases/LayerNorms are perturbed."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(state_dict):
        v = state_dict[k]
        if k.endswith("sampling_offsets.weight"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.01)
        elif k.endswith("sampling_offsets.bias"):
            continue
        elif k.endswith("attention_weights.weight"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
        elif k.endswith("attention_weights.bias"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.5)
        elif ".norms." in k:
            base = 1.0 if k.endswith("weight") else 0.0
            v.copy_(base + torch.randn(v.shape, generator=g) * 0.05)
        elif v.dim() == 2:
            bound = math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * bound)
        else:
            v.copy_(torch.randn(v.shape, generator=g) * 0.02)
    return state_dict


def make_msda_case(N, Q, M, D, shapes, P, seed=0, coherent=False, dtype=torch.float32):
    """Kernel-level inputs (value, spatial_shapes, level_start_index,
    sampling_locations, attention_weights) in the operator's layout
    (multi_scale_deformable_attn_function.py:97-112).  ``loc ~ U(-0.1, 1.1)``
    (about 17 % of taps per axis fall outside the map) or, with ``coherent``,
    a raster of reference points plus a few pixels of noise."""
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.tensor(shapes, dtype=torch.long).reshape(-1, 2)
    L = shapes_t.shape[0]
    S = int(shapes_t.prod(1).sum()) if L else 0
    start = torch.cat([shapes_t.new_zeros(1), shapes_t.prod(1).cumsum(0)[:-1]]) if L \
        else shapes_t.new_zeros(0)
    value = torch.randn(N, S, M, D, generator=g)
    if coherent:
        side = max(1, int(math.ceil(math.sqrt(max(Q, 1)))))
        qi = torch.arange(Q)
        ref = torch.stack([((qi % side) + 0.5) / side, ((qi // side) + 0.5) / side], -1)
        noise = torch.randn(N, Q, M, L, P, 2, generator=g)
        wh = shapes_t.flip(-1).to(torch.float32)
        loc = ref[None, :, None, None, None, :] + noise * 4.0 / wh[None, None, None, :, None, :]
    else:
        loc = torch.rand(N, Q, M, L, P, 2, generator=g) * 1.2 - 0.1
    attn = torch.randn(N, Q, M, L * P, generator=g).softmax(-1).reshape(N, Q, M, L, P)
    return value.to(dtype), shapes_t, start, loc, attn


def _project_numpy(name):
    """Pillar anchors of workload ``name`` projected into the six synthetic
    cameras (same geometry as encoder.py:88-149, float64 numpy; used only to
    build realistic kernel-level access patterns, not for parity)."""
    w = WORKLOADS[name]
    H, W, Dz = w["bev_h"], w["bev_w"], 4
    zs = (np.linspace(0.5, 8 - 0.5, Dz) / 8) * (PC_RANGE[5] - PC_RANGE[2]) + PC_RANGE[2]
    xs = ((np.arange(W) + 0.5) / W) * (PC_RANGE[3] - PC_RANGE[0]) + PC_RANGE[0]
    ys = ((np.arange(H) + 0.5) / H) * (PC_RANGE[4] - PC_RANGE[1]) + PC_RANGE[1]
    gy, gx = np.meshgrid(ys, xs, indexing="ij")
    pts = np.stack([np.broadcast_to(gx.reshape(-1, 1), (H * W, Dz)),
                    np.broadcast_to(gy.reshape(-1, 1), (H * W, Dz)),
                    np.broadcast_to(zs.reshape(1, -1), (H * W, Dz)),
                    np.ones((H * W, Dz))], -1)                      # (Q,Dz,4)
    mats = np.stack(camera_rig(w["s"]))                               # (Nc,4,4)
    cam = np.einsum("nij,qdj->nqdi", mats, pts)
    z = cam[..., 2]
    uv = cam[..., :2] / np.maximum(z, 1e-5)[..., None]
    uv[..., 0] /= w["img"][1]
    uv[..., 1] /= w["img"][0]
    mask = (z > 1e-5) & (uv[..., 0] > 0) & (uv[..., 0] < 1) & (uv[..., 1] > 0) & (uv[..., 1] < 1)
    return uv.astype(np.float32), mask


def make_sca_msda_case(name, seed=0, num_heads=8, num_points=8):
    """Operator-level inputs with the access pattern of the SCA call of
    workload ``name`` (spatial_cross_attention.py:136-164): per-camera visible
    queries in BEV raster order, 4 pillar anchors, head-direction offset grid
    (init of spatial_cross_attention.py:255-267) plus noise.  Every camera is
    padded to ``max_len`` rows with zero reference points, like the reference's
    rebatch.  Returns (value, shapes, start, loc, attn, hits_per_camera)."""
    w = WORKLOADS[name]
    uv, mask = _project_numpy(name)
    Nc, Q, Dz, _ = uv.shape
    idx = [np.nonzero(mask[i].any(-1))[0] for i in range(Nc)]
    max_len = max(len(i) for i in idx)
    ref = np.zeros((Nc, max_len, Dz, 2), np.float32)
    for i in range(Nc):
        ref[i, :len(idx[i])] = uv[i, idx[i]]
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(w["shapes"], dtype=torch.long)
    L = shapes.shape[0]
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    S = int(shapes.prod(1).sum())
    D = EMBED_DIMS // num_heads
    value = torch.randn(Nc, S, num_heads, D, generator=g)
    th = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = grid / grid.abs().max(-1, keepdim=True)[0]
    grid = grid.view(num_heads, 1, 1, 2).repeat(1, L, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    off = grid[None, None] + torch.randn(Nc, max_len, num_heads, L, num_points, 2, generator=g) * 0.5
    wh = shapes.flip(-1).float()
    off = off / wh[None, None, None, :, None, :]
    off = off.view(Nc, max_len, num_heads, L, num_points // Dz, Dz, 2)
    loc = torch.from_numpy(ref)[:, :, None, None, None, :, :] + off
    loc = loc.reshape(Nc, max_len, num_heads, L, num_points, 2).contiguous()
    attn = torch.randn(Nc, max_len, num_heads, L * num_points, generator=g).softmax(-1)
    attn = attn.view(Nc, max_len, num_heads, L, num_points).contiguous()
    return value, shapes, start, loc, attn, [len(i) for i in idx]


def make_tsa_msda_case(name, seed=0, num_heads=8, num_points=4):
    """Operator-level inputs with the access pattern of the TSA call
    (temporal_self_attention.py:203-249): N = 2 queue entries over the BEV grid,
    one level, reference point = own cell (entry 0 shifted by the ego motion)."""
    w = WORKLOADS[name]
    H, W = w["bev_h"], w["bev_w"]
    Q = H * W
    g = torch.Generator().manual_seed(seed)
    D = EMBED_DIMS // num_heads
    value = torch.randn(2, Q, num_heads, D, generator=g)
    shapes = torch.tensor([[H, W]], dtype=torch.long)
    start = torch.zeros(1, dtype=torch.long)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ref = torch.stack([(xs.flatten() + 0.5) / W, (ys.flatten() + 0.5) / H], -1)
    ref = torch.stack([ref + torch.tensor([0.0, 2.5 / 0.512 / H]), ref], 0)      # (2,Q,2)
    th = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = grid / grid.abs().max(-1, keepdim=True)[0]
    grid = grid.view(num_heads, 1, 1, 2).repeat(1, 1, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    off = grid[None, None] + torch.randn(2, Q, num_heads, 1, num_points, 2, generator=g) * 0.5
    loc = ref[:, :, None, None, None, :] + off / torch.tensor([W, H], dtype=torch.float32)
    attn = torch.randn(2, Q, num_heads, num_points, generator=g).softmax(-1)
    return value, shapes, start, loc.contiguous(), attn.view(2, Q, num_heads, 1, num_points).contiguous()
