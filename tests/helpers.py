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
