"""Shared test helpers (tests may use oracle/; the product package may not)."""
import contextlib

import torch

from bevformer_amd import ops
from oracle import bevformer_cpu as O


def _oracle_msda(value, shapes, start, loc, attn, im2col_step=64, tag=None):
    return O.msda_gridsample(value, shapes, loc, attn)


def _oracle_msda_ragged(value, shapes, start, loc, attn, row_batch, tag=None):
    """Ragged batch through the oracle: one call per value-batch entry."""
    R, M = loc.shape[:2]
    out = value.new_zeros(R, M * value.shape[-1])
    for n in range(value.shape[0]):
        sel = (row_batch == n).nonzero().squeeze(-1)
        if sel.numel():
            out[sel] = O.msda_gridsample(value[n:n + 1], shapes, loc[sel][None], attn[sel][None])[0]
    return out


def _oracle_msda_fused(value, shapes, start, proj, n_off, ref, row_batch, *, M, L, P, K, off_head,
                       off_k, lg_head, lg_k, ref_mode, vmul, vadd, Q=0, row_src=None, tag=None, **_lds):
    """CPU statement of the fused entry point's contract (include/bevmsda.h,
    ``bevmsda_fused_forward_*``) out of torch ops + the oracle operator: what the
    kernel must compute for a given descriptor."""
    if row_src is not None:                 # one projection row per BEV query, shared by its rows
        proj = proj[row_src.long()]
    R = proj.shape[0]
    D = value.shape[-1]
    A = ref.shape[-2]
    ref = ref.reshape(R, K, A, 2)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(proj.dtype)            # (L,2) = (W,H)
    m = torch.arange(M)[:, None]
    lp = torch.arange(L * P)[None, :]
    base = row_batch.long() if row_batch is not None else torch.arange(R) // Q
    out = value.new_zeros(R, M * D)
    for k in range(K):
        col_lg = n_off + m * lg_head + k * lg_k + lp                                 # (M, L*P)
        att = proj[:, col_lg].softmax(-1).view(R, M, L, P)
        col_off = (m * off_head + k * off_k + lp * 2)[..., None] + torch.arange(2)   # (M, L*P, 2)
        off = proj[:, col_off].view(R, M, L, P, 2)
        if ref_mode == 0:
            rp = ref[:, k][:, torch.arange(P) % A]                                   # (R,P,2)
            loc = rp[:, None, None, :, :] + off / norm[None, None, :, None, :]
        else:
            loc = ref[:, k][:, None, :, None, :] + off / norm[None, None, :, None, :]
        n = (base * vmul + k * vadd).to(torch.int32)
        out = out + _oracle_msda_ragged(value, shapes, start, loc, att, n)
    return out / K


def _oracle_gather_mean(rows, idx, scale):
    """Contract of ``bevmsda_gather_mean_f32`` in torch ops."""
    Qn, J = idx.shape
    out = rows.new_zeros(Qn, rows.shape[1])
    for j in range(J):
        sel = idx[:, j].long()
        ok = sel >= 0
        out[ok] += rows[sel[ok]]
    return out * scale.reshape(-1, 1)


def _oracle_rotate_bev(prev_bev, angles_deg, center, bev_h, bev_w):
    out = prev_bev.clone()
    for i in range(prev_bev.shape[1]):
        img = prev_bev[:, i].reshape(bev_h, bev_w, -1).permute(2, 0, 1)
        out[:, i] = O.rotate_nearest(img, angles_deg[i], center).permute(1, 2, 0).reshape(bev_h * bev_w, -1)
    return out


def kernel_rotation_index(h, w, angle_deg, center, device, device_pose=False):
    """The source-pixel map (h*w,) int64 (-1 = zero fill) the product's rotate kernel applies for this pose, read
    off an index image; ``device_pose`` takes the GraphedBevHistory route (angle in a device tensor, matrix from
    ``ops.rotation_theta_device``)."""
    img = (torch.arange(h * w, dtype=torch.float32) + 1).view(-1, 1, 1).expand(-1, 1, 256).contiguous().to(device)
    ang = torch.tensor([angle_deg], dtype=torch.float64, device=device) if device_pose else [float(angle_deg)]
    return ops.rotate_bev(img, ang, center, h, w)[:, 0, 0].round().long().cpu() - 1


def _oracle_flatten_feats(mlvl_feats, cams_embeds, level_embeds):
    """Contract of ``bevmsda_flatten_feats_f32`` in torch ops (transformer.py:165-184)."""
    flat, shapes = [], []
    for lvl, feat in enumerate(mlvl_feats):
        h, w = feat.shape[3:]
        feat = feat.flatten(3).permute(1, 0, 3, 2)
        if cams_embeds is not None:
            feat = feat + cams_embeds[:, None, None, :]
        feat = feat + level_embeds[None, None, lvl:lvl + 1, :]
        shapes.append((h, w))
        flat.append(feat)
    out = torch.cat(flat, 2).permute(0, 2, 1, 3).contiguous()
    ss = torch.as_tensor(shapes, dtype=torch.long)
    return out, ss, torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))


@contextlib.contextmanager
def oracle_ops():
    """Route the package's operator calls through the CPU oracle so that the
    HOST logic of the modules (ragged rows, merged GEMMs, geometry, plans,
    tiling) can be parity-tested without a GPU.  Test-only: the product path
    itself has no CPU implementation."""
    saved = (ops.msda, ops.msda_ragged, ops.msda_fused, ops.gather_mean, ops.rotate_bev,
             ops.flatten_feats)
    ops.msda, ops.msda_ragged, ops.msda_fused, ops.gather_mean, ops.rotate_bev, ops.flatten_feats = \
        _oracle_msda, _oracle_msda_ragged, _oracle_msda_fused, _oracle_gather_mean, \
        _oracle_rotate_bev, _oracle_flatten_feats
    try:
        yield
    finally:
        (ops.msda, ops.msda_ragged, ops.msda_fused, ops.gather_mean, ops.rotate_bev,
         ops.flatten_feats) = saved


def build_pair(name, seed=3, device="cpu"):
    """(product encoder, reference-keyed state_dict with 'trained-like' weights)."""
    import bevformer_amd
    from bevformer_amd import synthetic as S
    torch.manual_seed(0)
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg(name)).eval()
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    S.trained_like_(sd, seed=seed)
    enc.load_state_dict(sd)
    return enc.to(device), sd


def build_transformer_pair(name, seed=3, device="cpu"):
    """(product PerceptionTransformer, its state_dict with trained-like encoder weights and
    N(0,1)-scale embeddings / can-bus MLP)."""
    import bevformer_amd
    from bevformer_amd import synthetic as S
    torch.manual_seed(0)
    t = bevformer_amd.build_transformer(S.transformer_cfg(name)).eval()
    t.init_weights()
    sd = {k: v.clone() for k, v in t.state_dict().items()}
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    S.trained_like_(enc, seed=seed)
    for k, v in enc.items():
        sd["encoder." + k] = v
    t.load_state_dict(sd)
    return t.to(device), sd


def split_transformer_sd(sd):
    """-> (transformer-own parameters, encoder state_dict without the prefix)."""
    own = {k: v for k, v in sd.items() if not k.startswith(("encoder.", "decoder."))}
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    return own, enc


# ---------------------------------------------------------------------------------------------------------------------
# The oracle on a SUBSET of BEV queries, and the edge-adjacent sampling points of a frame
# ---------------------------------------------------------------------------------------------------------------------
def _tsa_rows(sd, pre, query, value, hist_rows, bev_pos, ref_2d, bev_h, bev_w, msda, num_heads=8, num_points=4):
    """``O.temporal_self_attention`` (temporal_self_attention.py:177-272) for the query rows ``query`` (bs, R, C) of a
    grid whose value tensor ``value`` (bs * 2, Q, C) is complete; ``hist_rows`` = value[:bs] at the same rows (the first
    half of the projection's input).  The only change against the oracle function: the rows of the concatenation."""
    bs, R, C = query.shape
    identity = query
    query = query + bev_pos
    shapes = torch.tensor([[bev_h, bev_w]])
    q2 = torch.cat([hist_rows, query], -1)
    v = O._lin(sd, pre + "value_proj", value).reshape(bs * 2, value.shape[1], num_heads, -1)
    off = O._lin(sd, pre + "sampling_offsets", q2).view(bs, R, num_heads, 2, 1, num_points, 2)
    att = O._lin(sd, pre + "attention_weights", q2).view(bs, R, num_heads, 2, num_points)
    att = att.softmax(-1).view(bs, R, num_heads, 2, 1, num_points)
    att = att.permute(0, 3, 1, 2, 4, 5).reshape(bs * 2, R, num_heads, 1, num_points).contiguous()
    off = off.permute(0, 3, 1, 2, 4, 5, 6).reshape(bs * 2, R, num_heads, 1, num_points, 2)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref_2d[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = msda(v, shapes, loc, att)
    out = out.permute(1, 2, 0).view(R, C, bs, 2).mean(-1).permute(2, 0, 1)
    return O._lin(sd, pre + "output_proj", out) + identity


def oracle_encoder_rows(sd, bev_query, feats, rows, *, bev_h, bev_w, bev_pos, spatial_shapes, prev_bev, shift, img_metas,
                        pc_range, msda=None, num_points_in_pillar=4, **_):
    """Rows ``rows`` (1-D long tensor of BEV query ids) of ``O.encoder_forward`` WITH a history BEV, out of the oracle's own
    pieces: every per-query operation of a layer is row-wise, the only cross-query reads are the sampling values
    (camera features; [history, layer-0 queries]), which stay complete.  Pinned against the full oracle on the CPU
    (tests/test_oracle.py::test_oracle_rows_helper_equals_the_full_oracle).  Any dtype (float64 for the gradient checks)."""
    msda = msda or O.msda_gridsample
    num_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    bs = bev_query.size(1)
    dt = bev_query.dtype
    # the geometry in float32 whatever ``dt``: the reference computes it in float32 (encoder.py:86, @force_fp32), and a
    # float64 projection would decide borderline visibility tests differently from every float32 implementation
    g = torch.float32
    ref_3d = O.pillar_points(bev_h, bev_w, pc_range[5] - pc_range[2], num_points_in_pillar, bs, g)
    ref_2d = O.bev_grid_points(bev_h, bev_w, bs, g)
    ref_cam, bev_mask = O.project_to_cameras(ref_3d, pc_range, img_metas)
    shifted = (ref_2d.clone() + shift.to(g)[:, None, None, :]).to(dt)
    ref_2d, ref_cam = ref_2d.to(dt), ref_cam.to(dt)
    x_full = bev_query.permute(1, 0, 2)
    pos = bev_pos.permute(1, 0, 2)[:, rows]
    Q = x_full.shape[1]
    prev = torch.stack([prev_bev.permute(1, 0, 2), x_full], 1).reshape(bs * 2, Q, -1)
    hybrid = torch.stack([shifted, ref_2d], 1).reshape(bs * 2, Q, 1, 2)[:, rows]
    hist_rows = prev[:bs][:, rows]
    ref_cam, bev_mask = ref_cam[:, :, rows], bev_mask[:, :, rows]
    x = x_full[:, rows]
    for i in range(num_layers):
        pre = f"layers.{i}."
        x = _tsa_rows(sd, pre + "attentions.0.", x, prev, hist_rows, pos, hybrid, bev_h, bev_w, msda)
        x = O.layer_norm(sd, pre + "norms.0", x)
        x = O.spatial_cross_attention(sd, pre + "attentions.1.", x, feats, ref_cam, bev_mask, spatial_shapes, msda=msda)
        x = O.layer_norm(sd, pre + "norms.1", x)
        h = O._lin(sd, pre + "ffns.0.layers.0.0", x)           # O.ffn, with the pre-activations shown to the recorder
        if hasattr(msda, "preactivations"):
            msda.preactivations(h)
        x = x + O._lin(sd, pre + "ffns.0.layers.1", torch.relu(h))
        x = O.layer_norm(sd, pre + "norms.2", x)
    return x


class EdgeRecorder:
    """An ``msda=`` stand-in for the oracle that evaluates the operator and REMEMBERS which BEV queries own a sampling point
    within ``eps`` pixels of a pixel boundary (x = loc_x * W - 0.5 within eps of an integer, or y): bilinear sampling is
    piecewise linear in the location, so such a point takes the slope of one cell in one evaluation and of its neighbour
    in another when round-off moves it across — a comparison of gradients means something on the OTHER queries.
    ``cam_rows[i]`` = the BEV query of every row of camera i's rebatch (``bev_mask[i][0].sum(-1).nonzero()``: the
    oracle's own list); calls with 2 * bs value entries are TemporalSelfAttention's (row = query)."""

    def __init__(self, num_queries, cam_rows, eps=1e-4, query_ids=None, relu_eps=0.0):
        self.fragile = torch.zeros(num_queries, dtype=torch.bool)
        self.cam_rows, self.eps, self.relu_eps = cam_rows, eps, relu_eps
        self.query_ids = query_ids          # rows of a subset evaluation -> BEV query ids (None: identity)

    def preactivations(self, h):
        """The FFN's ReLU is the other kink of a layer: rows with a pre-activation within ``relu_eps`` of zero."""
        if self.relu_eps:
            near = (h.detach().abs() < self.relu_eps).any(-1).any(0)
            ids = torch.arange(near.shape[0]) if self.query_ids is None else self.query_ids
            self.fragile[ids[near]] = True

    def __call__(self, value, shapes, loc, att):
        out = O.msda_gridsample(value, shapes, loc, att)
        with torch.no_grad():
            wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(loc.dtype)           # (L, 2) = (W, H)
            px = loc.detach() * wh[None, None, None, :, None, :] - 0.5
            fr = px - px.floor()
            near = (torch.minimum(fr, 1 - fr) < self.eps).any(-1).flatten(2).any(-1)    # (N, rows)
            if value.shape[0] == len(self.cam_rows):                                    # SCA: one value entry per camera
                for i, ids in enumerate(self.cam_rows):
                    self.fragile[ids[near[i, :len(ids)]]] = True
            else:
                ids = torch.arange(near.shape[1]) if self.query_ids is None else self.query_ids
                self.fragile[ids[near.any(0)]] = True
        return out


def camera_rows(name, rows=None):
    """The BEV query of every row of the oracle's per-camera rebatch for workload ``name`` (spatial_cross_attention.py:136-141),
    optionally for a subset evaluation over queries ``rows`` (ids are then BEV query ids of the subset's rows)."""
    from bevformer_amd import synthetic as S
    w = S.WORKLOADS[name]
    ref_3d = O.pillar_points(w["bev_h"], w["bev_w"], S.PC_RANGE[5] - S.PC_RANGE[2], 4, 1, torch.float32)
    _, bev_mask = O.project_to_cameras(ref_3d, S.PC_RANGE, S.make_img_metas(name))
    if rows is not None:
        bev_mask = bev_mask[:, :, rows]
    out = []
    for m in bev_mask:
        local = m[0].sum(-1).nonzero().squeeze(-1)
        out.append(local if rows is None else rows[local])
    return out
