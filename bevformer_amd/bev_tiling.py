"""BEV-query tiling of ONE frame over the GPUs of a node.

The reference only knows data parallelism (one sample per GPU,
bevformer/apis/mmdet_train.py:75-79).  This module adds the partition named by
the north star: the ``bev_h x bev_w`` grid is cut into tiles, rank r runs the
whole layer stack on its tile, and the grid is reassembled with an all-gather
(RCCL over xGMI; ``gloo`` in the CPU tests).  Two layouts: ``rows`` —
contiguous blocks of BEV rows — and ``sectors`` — equal ranges of the cells
ordered by azimuth around the ego vehicle (``geometry.sector_permutation``): a
sector is seen by 1-3 of the 6 cameras where a row block is seen by 3-4, and
the replicated camera-value projection skips the cameras a rank cannot see.
In the sector layout the per-query tensors of a rank (queries, positional
encoding, frame plan, reference points) are in sector order; everything that is
SAMPLED spatially (camera features, the history / current BEV grid) and the
encoder's output stay in grid order.

Why this is exact: every per-query operation of a layer (projections, softmax,
sampling, scatter-mean, output projection, FFN, LayerNorm; encoder.py:356-404)
is row-wise in the BEV queries.  The only cross-query reads are the sampling
*values*:
  * SCA samples the camera features  -> replicated input, no exchange;
  * TSA with history samples ``[prev_bev, layer-0 bev_query]`` which the
    reference builds ONCE before the layer loop (encoder.py:204-209,229)
    -> replicated input, no per-layer exchange;
  * TSA without history samples the *current* layer input
    (temporal_self_attention.py:177-180) -> one all-gather per layer.
One all-gather at the exit reassembles the output (it is the next frame's
``prev_bev`` and the decoder's value).

Replicated compute (Amdahl terms, measured in DESIGN.md): the value projections
of SCA (camera features) and TSA (full BEV) run on every rank.
"""
from dataclasses import dataclass, replace
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class BevTiling:
    """``simulate = (rank, world)``: run that rank's schedule of a ``world``-GPU job in ONE process without a
    process group — every kernel of the rank's step is launched, the all-gather is replaced by the copy of
    the rank's own shard into the full grid (the other rows stay zero).  For per-rank timing on a single
    GPU (bench.py's ``multi_gpu_model``); the output is NOT the encoder's output."""
    group: Optional[object] = None
    simulate: Optional[tuple] = None
    layout: str = "rows"                 # "rows" | "sectors" (enable_bev_tiling resolves "auto")

    @property
    def world(self):
        return self.simulate[1] if self.simulate else dist.get_world_size(self.group)

    @property
    def rank(self):
        return self.simulate[0] if self.simulate else dist.get_rank(self.group)


LAYOUTS = ("auto", "rows", "sectors")


def enable_bev_tiling(encoder, group=None, simulate=None, layout="auto"):
    """Switch ``encoder.forward`` to the tiled schedule on an initialised
    ``torch.distributed`` process group (one process per GPU); ``simulate``: see ``BevTiling``;
    ``layout``: ``rows``, ``sectors`` (module docstring) or ``auto`` = by measurement (bench.py ``multi_gpu_model``,
    profiles/r3): two half-planes see as many cameras as two row blocks and sample less locally (0.65 vs 0.68 modelled
    efficiency), from 3 ranks on the sectors win (8 ranks: 2 cameras per rank instead of 3-4, 0.32 vs 0.30)."""
    if simulate is None and not dist.is_initialized():
        raise RuntimeError("enable_bev_tiling needs an initialised torch.distributed group")
    if layout not in LAYOUTS:
        raise ValueError(f"layout must be one of {LAYOUTS}")
    t = BevTiling(group, tuple(simulate) if simulate is not None else None, layout)
    if layout == "auto":
        t.layout = "sectors" if t.world >= 3 else "rows"
    encoder.bev_tiling = t
    return encoder


def disable_bev_tiling(encoder):
    encoder.bev_tiling = None
    return encoder


def row_blocks(bev_h, world):
    """Contiguous split of the BEV rows; the first ``bev_h % world`` ranks get
    one extra row.  -> list of (h0, h1)."""
    base, extra = divmod(bev_h, world)
    out, h = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((h, h + n))
        h += n
    return out


def query_blocks(num_queries, world):
    """Equal contiguous ranges of a query order (the sector layout): -> list of (q0, q1)."""
    base, extra = divmod(num_queries, world)
    out, q = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((q, q + n))
        q += n
    return out


_PERMS = {}                      # for callers without an encoder (tests, tools); guarded by _PERMS_LOCK
_PERMS_LOCK = __import__("threading").Lock()


def sector_order(bev_h, bev_w, pc_range, device, group=None, collective=False, cache=None):
    """(name, perm, inverse, perm on the CPU) of the sector layout on ``device``: ``perm[q'] = cell``,
    ``inverse[cell] = q'``.  ``collective`` (a real process group, not a simulated rank): every rank takes RANK 0's
    permutation (one broadcast, once per grid) — the order comes out of float ``arctan2`` / ``hypot`` and a sort, and
    ranks whose libm or numpy round a tie differently would otherwise slice their tiles from different orders and
    reassemble a silently scrambled grid."""
    key = (bev_h, bev_w, tuple(float(v) for v in pc_range), str(device), id(group) if collective else None)
    if cache is None:               # ``cache``: the encoder's own table (``tiled_forward``) — no state shared between encoders
        with _PERMS_LOCK:
            return sector_order(bev_h, bev_w, pc_range, device, group, collective, cache=_PERMS)
    hit = cache.get(key)
    if hit is None:
        from .modules.geometry import sector_permutation
        perm = sector_permutation(bev_h, bev_w, [float(v) for v in pc_range])
        if collective and dist.is_initialized() and dist.get_world_size(group) > 1:
            shared = perm.to(device)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast(shared, src=src, group=group)
            perm = shared.cpu()
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        hit = cache[key] = (f"sectors{bev_h}x{bev_w}", perm.to(device), inv.to(device), perm)
    return hit


def slice_plan(plan, q0, q1):
    """The frame plan restricted to BEV queries [q0, q1) of every batch entry."""
    Q = plan.bev_h * plan.bev_w
    b = plan.row_query // Q
    q = plan.row_query - b * Q
    keep = (q >= q0) & (q < q1)
    new_row_query = (b[keep] * (q1 - q0) + (q[keep] - q0)).contiguous()
    from .modules.geometry import build_q_rows, camera_runs
    new_row_batch = plan.row_batch[keep].contiguous()
    cam_start, max_cam_rows = camera_runs(new_row_batch, plan.cam_start.numel() - 1) \
        if plan.cam_start is not None else (None, 0)
    return replace(
        plan, ref_3d=plan.ref_3d[:, :, q0:q1], ref_2d=plan.ref_2d[:, q0:q1],
        reference_points_cam=plan.reference_points_cam[:, :, q0:q1],
        bev_mask=plan.bev_mask[:, :, q0:q1],
        row_query=new_row_query, row_query32=new_row_query.to(torch.int32),
        q_rows=build_q_rows(new_row_query, plan.bs * (q1 - q0)),
        row_batch=new_row_batch, row_ref=plan.row_ref[keep].contiguous(),
        inv_count=plan.inv_count[:, q0:q1].contiguous(), hits=[], cam_start=cam_start,
        max_cam_rows=max_cam_rows)


def all_gather_rows(local, blocks, bev_w, group=None, simulate=None, inverse=None):
    """local (bs, rows_r*bev_w, C) on every rank -> (bs, Q, C).  Blocks may be
    uneven (padded to the largest for the collective).  ``inverse`` (sector layout): the gathered queries are in
    sector order; ``inverse[cell]`` is the position of a grid cell in it — the result is in grid order."""
    world = len(blocks)
    sizes = [(h1 - h0) * bev_w for h0, h1 in blocks]
    if inverse is not None:
        return all_gather_rows(local, blocks, bev_w, group, simulate).index_select(1, inverse)
    if simulate is not None:                 # single-process timing run: my shard into an otherwise empty grid
        full = local.new_zeros(local.shape[0], sum(sizes), local.shape[2])
        q0 = sum(sizes[:simulate[0]])
        full[:, q0:q0 + sizes[simulate[0]]] = local
        return full
    mx = max(sizes)
    bs, n, C = local.shape
    if n < mx:
        local = torch.cat([local, local.new_zeros(bs, mx - n, C)], 1)
    # output is the dim-0 concatenation of the shards (the layout both RCCL and
    # gloo accept for all_gather_into_tensor), viewed back as (world, bs, mx, C)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # FUNCTIONAL path only (several ranks sharing one GPU in the tests and in `bench.py --dist-backend gloo`, where
        # RCCL refuses duplicate devices): gloo has no all-gather of device tensors, so the shards travel through host
        # memory.  Never a measured configuration — the product collective is the RCCL call below.
        host = torch.empty(world * bs, mx, C, dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        flat = host.to(local.device)
    else:
        flat = local.new_empty(world * bs, mx, C)
        dist.all_gather_into_tensor(flat, local.contiguous(), group=group)
    buf = flat.view(world, bs, mx, C)
    if all(s == mx for s in sizes):
        return buf.permute(1, 0, 2, 3).reshape(bs, world * mx, C)
    return torch.cat([buf[r, :, :sizes[r]] for r in range(world)], 1)


def rank_tile(encoder, bev_h, bev_w, device):
    """This rank's share of the queries under ``encoder.bev_tiling``: ``(blocks, unit, (q0, q1), cell_perm, rows_idx,
    inverse)`` — the blocks of every rank in units of ``unit`` queries, my range of the (row- or sector-ordered) queries, the
    sector permutation handed to the frame plan (None for row tiles), the grid cells of my queries and the grid-order
    inverse of the sector order (both None for row tiles)."""
    tiling = encoder.bev_tiling
    group, world, rank = tiling.group, tiling.world, tiling.rank
    if tiling.layout == "sectors":
        # queries in sector order: tile = a contiguous range of that order; blocks in units of ONE query
        pname, perm, inverse, perm_cpu = sector_order(bev_h, bev_w, encoder.pc_range, device, group=group,
                                                      collective=tiling.simulate is None,
                                                      cache=encoder.__dict__.setdefault("_sector_orders", {}))
        blocks, unit = query_blocks(bev_h * bev_w, world), 1
        q0, q1 = blocks[rank]
        return blocks, unit, (q0, q1), (pname, perm_cpu), perm[q0:q1], inverse
    blocks, unit = row_blocks(bev_h, world), bev_w
    h0, h1 = blocks[rank]
    return blocks, unit, (h0 * bev_w, h1 * bev_w), None, None, None


def tiled_forward(encoder, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                  spatial_shapes=None, level_start_index=None, prev_bev=None, shift=0.0,
                  **kwargs):
    """Same contract as ``BEVFormerEncoder.forward``; every rank returns the
    full (bs, Q, C) grid."""
    from . import ops
    from .modules import geometry
    tiling = encoder.bev_tiling
    group, world, rank = tiling.group, tiling.world, tiling.rank
    bs = bev_query.size(1)
    if torch.is_grad_enabled() and any(p.requires_grad for p in encoder.parameters()):
        # all_gather_into_tensor is not differentiable: the tiled schedule is inference-only
        raise RuntimeError("BEV tiling is an inference schedule (its all-gather has no autograd); "
                           "call it under torch.no_grad() or disable_bev_tiling() for training")
    sectors = tiling.layout == "sectors"
    blocks, unit, (q0, q1), cell_perm, rows_idx, inverse = rank_tile(encoder, bev_h, bev_w, bev_query.device)
    plan_ready = None
    if encoder.device_plans and bev_query.is_cuda:
        # device-side plan of my tile: rows only for queries [q0, q1), tile-local slot numbering
        overlap = ops.modes().overlap_value_proj if encoder.overlap_value_proj is None else encoder.overlap_value_proj
        if overlap and ops.modes().plan_on_side and value.is_cuda and not ops.gemm_timer_active():
            # (as BEVFormerEncoder._forward: the plan kernels ahead of the camera-value projection on ITS stream — that
            # projection reads the plan's camera segments — and the main stream meets both at the first SpatialCrossAttention)
            cur = torch.cuda.current_stream(value.device)
            if encoder._side_stream is None or encoder._side_stream.device != value.device:
                encoder._side_stream = torch.cuda.Stream(value.device)
            encoder._side_stream.wait_stream(cur)
            with torch.cuda.stream(encoder._side_stream):
                tile = encoder.frame_plan(bev_h, bev_w, bs, kwargs["img_metas"], bev_query.device,
                                          bev_query.dtype, tile=(q0, q1), cell_perm=cell_perm)
                plan_ready = encoder._side_stream.record_event()
        else:
            tile = encoder.frame_plan(bev_h, bev_w, bs, kwargs["img_metas"], bev_query.device,
                                      bev_query.dtype, tile=(q0, q1), cell_perm=cell_perm)
        full_ref_2d = tile.ref_2d_full
    else:
        plan = encoder.frame_plan(bev_h, bev_w, bs, kwargs["img_metas"], bev_query.device,
                                  bev_query.dtype, cell_perm=cell_perm)
        full_ref_2d = plan.ref_2d
        cache = getattr(plan, "_tiles", None)
        if cache is None:
            cache = plan._tiles = {}
        tile = cache.get((q0, q1))
        if tile is None:
            tile = cache[(q0, q1)] = slice_plan(plan, q0, q1)

    ref_2d = full_ref_2d
    full_query = bev_query.permute(1, 0, 2)
    take = (lambda t: t.index_select(1, rows_idx)) if sectors else (lambda t: t[:, q0:q1])
    pos_local = take(bev_pos.permute(1, 0, 2)).contiguous()     # (once per frame, not once per layer)
    Q = ref_2d.shape[1]
    stack_free = False
    if prev_bev is not None:
        history = prev_bev.permute(1, 0, 2)
        # (inference, bs = 1: the value projection reads history and queries where they lie — encoder.py, ``_stack_free``)
        stack_free = encoder._stack_free(history, full_query, bs)
        tsa_value = (history, full_query) if stack_free else torch.stack([history, full_query], 1).reshape(bs * 2, Q, -1)
        # (my queries' anchors only: stack([ref + shift, ref], 1)[:, q0:q1], one launch — geometry.hybrid_ref_2d)
        hybrid = geometry.hybrid_ref_2d(ref_2d[:, q0:q1], shift)
    else:
        tsa_value = None
        hybrid = torch.stack([ref_2d, ref_2d], 1).reshape(bs * 2, Q, 1, 2)[:, q0:q1].contiguous()

    x = take(full_query).contiguous()
    history_local = take(prev_bev.permute(1, 0, 2)).contiguous() if (prev_bev is not None and bs == 1) else None
    inter = []
    # replicated, layer-invariant value projections: one grouped GEMM each (encoder.py docstring)
    # (the tile's plan tells the camera-value projection which cameras this rank's queries can see at all)
    sca_vals, tsa_vals = encoder.hoisted_value_projections(value, tsa_value, plan=tile if world > 1 else None,
                                                            spatial_shapes=spatial_shapes)
    if plan_ready is not None and getattr(encoder, "_sca_ready", None) is None:
        torch.cuda.current_stream(value.device).wait_event(plan_ready)      # (no side-stream projection to meet: join now)
    if stack_free:
        tsa_value = history.expand(2, Q, history.shape[-1]) if tsa_vals is not None \
            else torch.stack([history, full_query], 1).reshape(bs * 2, Q, -1)
    seam = None
    for li, layer in enumerate(encoder.layers):
        hoisted = {}
        if seam is not None and seam.get("proj") is not None:
            hoisted["tsa_proj"] = seam["proj"]           # (the previous layer's last kernel made this layer's TSA projection)
        seam = encoder.tsa_seam(li, history_local, pos_local) if history_local is not None else None
        if seam is not None:
            hoisted["tsa_seam"] = seam
        if history_local is not None:
            hoisted["tsa_history"] = history_local       # (the K source of TSA's projection: gathered once, not per layer)
        if sca_vals is not None:
            hoisted["projected_value"] = sca_vals[li]
            if li == 0 and getattr(encoder, "_sca_ready", None) is not None:
                hoisted["projected_value_ready"] = encoder._sca_ready      # projection issued on a side stream
        if tsa_vals is not None:
            hoisted["tsa_projected_value"] = tsa_vals[li]
        if prev_bev is None:
            # no history: TSA's value is the CURRENT full BEV -> exchange per layer
            full = full_query if li == 0 else all_gather_rows(x, blocks, unit, group, tiling.simulate, inverse)
            layer_value = torch.stack([full, full], 1).reshape(bs * 2, Q, -1)
        else:
            layer_value = tsa_value
        x = layer(x, key, value, *args, bev_pos=pos_local, ref_2d=hybrid, ref_3d=tile.ref_3d,
                  bev_h=bev_h, bev_w=bev_w, spatial_shapes=spatial_shapes,
                  level_start_index=level_start_index,
                  reference_points_cam=tile.reference_points_cam, bev_mask=tile.bev_mask,
                  prev_bev=layer_value, frame_plan=tile, bev_slice=(q0, q1), bev_rows=rows_idx, **hoisted, **kwargs)
        if encoder.return_intermediate:
            inter.append(all_gather_rows(x, blocks, unit, group, tiling.simulate, inverse))
    if encoder.return_intermediate:
        return torch.stack(inter)
    return all_gather_rows(x, blocks, unit, group, tiling.simulate, inverse)
