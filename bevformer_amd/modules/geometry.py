"""Per-frame geometry of the BEV encoder, computed once and shared by all layers.

The reference recomputes the per-camera visibility lists inside every
SpatialCrossAttention call (``nonzero()`` + Python rebatch loops,
spatial_cross_attention.py:136-153: six host syncs per layer) and rebuilds tiny
device tensors per layer (encoder.py:370-372).  Here everything that depends
only on (BEV grid, camera matrices, image size) lives in a ``FramePlan``:

  * pillar / BEV reference points          (encoder.py:46-85)
  * camera projection + visibility mask    (encoder.py:88-149, fp32)
  * the ragged row list of SCA: for every (batch j, camera i) the BEV queries
    camera i sees, concatenated — ``row_query`` / ``row_batch`` / ``row_ref`` —
    instead of the reference's zero-padded (bs, num_cams, max_len) rebatch;
    rows the reference pads with zeros are never read back there
    (spatial_cross_attention.py:165-167), so results are identical
  * 1 / (number of cameras that see a query)   (spatial_cross_attention.py:169-172)

Plans are cached on the encoder keyed by the camera matrices, so a steady-state
frame performs no host<->device synchronisation at all.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch


def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim="3d", bs=1, device="cuda",
                         dtype=torch.float):
    """Same values and shapes as ``BEVFormerEncoder.get_reference_points``
    (encoder.py:46-85): '3d' -> (bs, D, H*W, 3), '2d' -> (bs, H*W, 1, 2)."""
    if dim == "3d":
        zs = torch.linspace(0.5, Z - 0.5, num_points_in_pillar, dtype=dtype, device=device) / Z
        xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device) / W
        ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device) / H
        D = num_points_in_pillar
        ref = torch.stack((xs.view(1, 1, W).expand(D, H, W), ys.view(1, H, 1).expand(D, H, W),
                           zs.view(D, 1, 1).expand(D, H, W)), -1)
        return ref.reshape(D, H * W, 3)[None].repeat(bs, 1, 1, 1)
    if dim == "2d":
        xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device) / W
        ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device) / H
        ref = torch.stack((xs.view(1, W).expand(H, W), ys.view(H, 1).expand(H, W)), -1)
        return ref.reshape(1, H * W, 1, 2).repeat(bs, 1, 1, 1)
    raise ValueError(dim)


def point_sampling(reference_points, pc_range, img_metas):
    """Project pillar anchors into every camera (encoder.py:95-144), fp32.

    Returns ``reference_points_cam`` (Nc, B, Q, D, 2) and ``bev_mask``
    (Nc, B, Q, D) like the reference, without materialising the
    (D, B, Nc, Q, 4, 4) repeated matrices."""
    l2i = np.asarray([m["lidar2img"] for m in img_metas])
    l2i = reference_points.new_tensor(l2i).float()                      # (B,Nc,4,4)
    p = reference_points.float()
    x = p[..., 0] * (pc_range[3] - pc_range[0]) + pc_range[0]           # (B,D,Q)
    y = p[..., 1] * (pc_range[4] - pc_range[1]) + pc_range[1]
    z = p[..., 2] * (pc_range[5] - pc_range[2]) + pc_range[2]
    m = l2i[:, :, None, None]                                            # (B,Nc,1,1,4,4)
    x, y, z = x[:, None], y[:, None], z[:, None]                         # (B,1,D,Q)
    cam = [m[..., r, 0] * x + m[..., r, 1] * y + m[..., r, 2] * z + m[..., r, 3]
           for r in range(3)]                                            # each (B,Nc,D,Q)
    eps = 1e-5
    depth = cam[2]
    mask = depth > eps
    denom = torch.clamp(depth, min=eps)
    u = cam[0] / denom / img_metas[0]["img_shape"][0][1]
    v = cam[1] / denom / img_metas[0]["img_shape"][0][0]
    mask = mask & (v > 0.0) & (v < 1.0) & (u < 1.0) & (u > 0.0)
    ref_cam = torch.stack((u, v), -1).permute(1, 0, 3, 2, 4).contiguous()  # (Nc,B,Q,D,2)
    bev_mask = mask.permute(1, 0, 3, 2).contiguous()                        # (Nc,B,Q,D)
    return ref_cam, bev_mask


@dataclass
class FramePlan:
    bs: int
    bev_h: int
    bev_w: int
    ref_3d: torch.Tensor
    ref_2d: torch.Tensor
    reference_points_cam: torch.Tensor
    bev_mask: torch.Tensor
    bev_shapes: torch.Tensor          # (1,2) int64 on device
    bev_start: torch.Tensor           # (1,)  int64 on device
    # ragged SCA rows
    row_query: Optional[torch.Tensor] = None   # (R,) int64: j*Q + q  (flat slot index)
    row_batch: Optional[torch.Tensor] = None   # (R,) int32: j*Nc + cam (value batch entry)
    row_ref: Optional[torch.Tensor] = None     # (R,D,2) projected anchors of the row
    inv_count: Optional[torch.Tensor] = None   # (bs,Q,1)
    row_query32: Optional[torch.Tensor] = None  # (R,) int32 copy of row_query (kernel-side index)
    q_rows: Optional[torch.Tensor] = None       # (bs*Q, J) int32: rows of every slot, -1 = none
    hits: List[int] = field(default_factory=list)
    level_shapes_host: Optional[list] = None
    # rows are grouped by value batch entry (j*Nc + cam): first row of every entry's run, for the
    # SCA kernel variant that stages one (camera, head) level slice in LDS
    cam_start: Optional[torch.Tensor] = None    # (bs*Nc + 1,) int32
    max_cam_rows: int = 0


def _morton_key(u, v, bits=7):
    """Interleave the top ``bits`` bits of u, v in [0, 1) -> int64 Z-order key."""
    n = 1 << bits
    ui = (u.clamp(0.0, 1.0) * (n - 1)).long()
    vi = (v.clamp(0.0, 1.0) * (n - 1)).long()
    key = torch.zeros_like(ui)
    for b in range(bits):
        key |= ((ui >> b) & 1) << (2 * b)
        key |= ((vi >> b) & 1) << (2 * b + 1)
    return key


ROW_ORDERS = ("raster", "image")


def build_sca_rows(reference_points_cam, bev_mask, row_order="raster"):
    """Ragged equivalent of spatial_cross_attention.py:136-153.

    Literal to the reference for bs > 1: the visible set of camera i is taken
    from batch element 0 and reused for every batch element; the camera count
    uses each element's own mask.

    ``row_order`` only permutes the rows inside a camera (results are
    unchanged: every row is scattered back through ``row_query``):
      * ``raster``: BEV raster order, as the reference's ``nonzero()`` yields;
      * ``image``:  Z-order of the projected pillar in the camera image, so that
        rows processed together sample neighbouring feature pixels."""
    assert row_order in ROW_ORDERS
    Nc, B, Q, D = bev_mask.shape
    vis0 = bev_mask[:, 0].any(-1)                                        # (Nc,Q)
    cam, q = vis0.nonzero(as_tuple=True)                                 # sorted by cam, then q (sync)
    hits = torch.bincount(cam, minlength=Nc).tolist()
    if row_order == "image" and cam.numel():
        m = bev_mask[cam, 0, q].float()                                  # (R0,D)
        uv = reference_points_cam[cam, 0, q]                             # (R0,D,2)
        cnt = m.sum(-1).clamp(min=1.0)
        u = (uv[..., 0] * m).sum(-1) / cnt
        v = (uv[..., 1] * m).sum(-1) / cnt
        key = cam.long() * (1 << 14) + _morton_key(u, v)
        perm = torch.argsort(key, stable=True)
        cam, q = cam[perm], q[perm]
    rows_q, rows_b, rows_ref = [], [], []
    for j in range(B):
        rows_q.append(j * Q + q)
        rows_b.append((j * Nc + cam).to(torch.int32))
        rows_ref.append(reference_points_cam[cam, j, q])                 # (R0,D,2)
    count = bev_mask.any(-1).permute(1, 2, 0).sum(-1)                    # (B,Q)
    inv_count = 1.0 / torch.clamp(count.float(), min=1.0)
    return (torch.cat(rows_q), torch.cat(rows_b).contiguous(), torch.cat(rows_ref).contiguous(),
            inv_count[..., None], hits)


def build_q_rows(row_query, num_slots):
    """Inverse of ``row_query``: for every slot (j*Q + q) the ids of the rows that
    scatter into it, in increasing row order, padded with -1 -> (num_slots, J) int32."""
    R = row_query.numel()
    counts = torch.bincount(row_query, minlength=num_slots)
    J = max(int(counts.max().item()) if R else 0, 1)
    table = torch.full((num_slots, J), -1, dtype=torch.int32, device=row_query.device)
    if R:
        order = torch.argsort(row_query, stable=True)
        sq = row_query[order]
        start = torch.cumsum(counts, 0) - counts
        pos = torch.arange(R, device=row_query.device) - start[sq]
        table[sq, pos] = order.to(torch.int32)
    return table


def build_frame_plan(bev_h, bev_w, bs, pc_range, num_points_in_pillar, img_metas, device,
                     dtype=torch.float32, row_order="raster"):
    ref_3d = get_reference_points(bev_h, bev_w, pc_range[5] - pc_range[2], num_points_in_pillar,
                                  dim="3d", bs=bs, device=device, dtype=dtype)
    ref_2d = get_reference_points(bev_h, bev_w, dim="2d", bs=bs, device=device, dtype=dtype)
    ref_cam, bev_mask = point_sampling(ref_3d, pc_range, img_metas)
    plan = FramePlan(bs=bs, bev_h=bev_h, bev_w=bev_w, ref_3d=ref_3d, ref_2d=ref_2d,
                     reference_points_cam=ref_cam, bev_mask=bev_mask,
                     bev_shapes=torch.tensor([[bev_h, bev_w]], device=device),
                     bev_start=torch.zeros(1, dtype=torch.long, device=device))
    (plan.row_query, plan.row_batch, plan.row_ref, plan.inv_count, plan.hits) = \
        build_sca_rows(ref_cam, bev_mask, row_order)
    plan.row_query32 = plan.row_query.to(torch.int32)
    plan.q_rows = build_q_rows(plan.row_query, bs * bev_h * bev_w)
    plan.cam_start, plan.max_cam_rows = camera_runs(plan.row_batch, bs * len(plan.hits))
    return plan


def camera_runs(row_batch, num_entries):
    """(first row of every value batch entry's run (num_entries + 1,) int32, longest run) of a
    row list grouped by entry (one host sync, at plan time)."""
    counts = torch.bincount(row_batch.long(), minlength=num_entries)
    start = torch.zeros(num_entries + 1, dtype=torch.int32, device=row_batch.device)
    start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return start, (int(counts.max().item()) if row_batch.numel() else 0)


def plan_key(bev_h, bev_w, bs, pc_range, num_points_in_pillar, img_metas, device, dtype):
    l2i = np.asarray([m["lidar2img"] for m in img_metas], dtype=np.float64)
    shp = tuple(tuple(int(v) for v in m["img_shape"][0][:2]) for m in img_metas[:1])
    return (bev_h, bev_w, bs, tuple(pc_range), num_points_in_pillar, l2i.tobytes(), shp,
            str(device), str(dtype))
