"""GPU: the device-side frame plan (csrc/frame_plan.h, ``bevmsda_frame_plan_f32``) against the
torch statements of the same geometry evaluated on the CPU (``geometry.build_frame_plan``, itself
parity-tested against the oracle / the reference in tests/test_host_logic_cpu.py and
tests/test_oracle_vs_reference.py), and the encoder running on device plans against the oracle.

Bar: index work (rows, row tables, masks, counts) bit-exact; projected coordinates bit-exact
(same IEEE operations in the same order — no FMA contraction in the kernel)."""
import math

import numpy as np
import pytest
import torch

from bevformer_amd import synthetic as S
from bevformer_amd.modules import geometry as G
from oracle import bevformer_cpu as O

from helpers import build_pair

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _perturbed_metas(name, bs, seed):
    """img_metas whose camera matrices differ per batch element (ego pose jitter)."""
    metas = S.make_img_metas(name, bs=bs)
    rng = np.random.default_rng(seed)
    for j, m in enumerate(metas):
        if j == 0 and seed == 0:
            continue
        yaw = rng.normal(0, 0.02)
        T = np.eye(4)
        T[:2, :2] = [[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]]
        T[:3, 3] = rng.normal(0, 0.3, 3)
        m["lidar2img"] = [mat @ T for mat in m["lidar2img"]]
    return metas


def _host_plan(name, bs, metas, order="raster"):
    w = S.WORKLOADS[name]
    return G.build_frame_plan(w["bev_h"], w["bev_w"], bs, S.PC_RANGE, 4, metas, "cpu", torch.float32,
                              row_order=order)


def _device_plan(name, bs, metas, order="raster", tile=None, cap=None):
    w = S.WORKLOADS[name]
    pl = G.DevicePlanner(w["bev_h"], w["bev_w"], bs, S.PC_RANGE, 4, S.NUM_CAMS, DEV, row_order=order,
                         tile=tile, row_capacity=cap)
    return pl, pl.plan(metas)


@pytest.mark.parametrize("name,bs", [("micro", 1), ("micro4", 2), ("tiny", 1), ("base", 1)])
def test_device_plan_equals_host_plan_raster(name, bs):
    metas = _perturbed_metas(name, bs, seed=1 if bs > 1 else 0)
    want = _host_plan(name, bs, metas)
    _, dyn = _device_plan(name, bs, metas)
    got = dyn.materialize()
    assert torch.equal(got.bev_mask.cpu(), want.bev_mask)
    assert torch.equal(got.reference_points_cam.cpu(), want.reference_points_cam), \
        (got.reference_points_cam.cpu() - want.reference_points_cam).abs().max()
    assert torch.equal(got.inv_count.cpu(), want.inv_count)
    assert got.hits == want.hits
    assert torch.equal(got.row_query.cpu(), want.row_query)
    assert torch.equal(got.row_batch.cpu(), want.row_batch)
    assert torch.equal(got.row_ref.cpu(), want.row_ref)
    J = want.q_rows.shape[1]
    assert torch.equal(got.q_rows.cpu()[:, :J], want.q_rows)
    assert (got.q_rows.cpu()[:, J:] == -1).all()
    assert torch.equal(got.cam_start.cpu()[:want.cam_start.numel()], want.cam_start)
    assert int(dyn.nrows_dev.item()) == want.row_query.numel()


@pytest.mark.parametrize("name", ["micro4", "tiny", "base"])
def test_device_plan_polar_order_is_a_permutation_inside_cameras(name):
    metas = S.make_img_metas(name)
    want = _host_plan(name, 1, metas)
    _, dyn = _device_plan(name, 1, metas, order="polar")
    got = dyn.materialize()
    assert got.hits == want.hits
    assert torch.equal(got.row_batch.cpu(), want.row_batch)           # rows stay grouped by camera
    key_w = want.row_batch.long() * 10 ** 6 + want.row_query
    key_g = got.row_batch.cpu().long() * 10 ** 6 + got.row_query.cpu()
    assert torch.equal(key_g.sort().values, key_w)                    # same (camera, query) pairs
    # each row carries its own anchors and the row table points back at it
    rq, rb = got.row_query.cpu(), got.row_batch.cpu().long()
    assert torch.equal(got.row_ref.cpu(), want.reference_points_cam[rb, 0, rq])
    qr = got.q_rows.cpu()
    for col in range(qr.shape[1]):
        sel = qr[:, col] >= 0
        assert torch.equal(rq[qr[sel, col].long()], torch.nonzero(sel).squeeze(-1))


@pytest.mark.parametrize("q0,q1", [(0, 60), (60, 120), (30, 90)])
def test_device_plan_of_a_tile_equals_the_sliced_host_plan(q0, q1):
    from bevformer_amd.bev_tiling import slice_plan
    metas = _perturbed_metas("micro4", 2, seed=3)
    want = slice_plan(_host_plan("micro4", 2, metas), q0, q1)
    _, dyn = _device_plan("micro4", 2, metas, tile=(q0, q1))
    got = dyn.materialize()
    assert torch.equal(got.row_query.cpu(), want.row_query)
    assert torch.equal(got.row_batch.cpu(), want.row_batch)
    assert torch.equal(got.row_ref.cpu(), want.row_ref)
    assert torch.equal(got.inv_count.cpu(), want.inv_count)
    J = want.q_rows.shape[1]
    assert torch.equal(got.q_rows.cpu()[:, :J], want.q_rows)
    assert torch.equal(got.bev_mask.cpu(), want.bev_mask)


def test_row_capacity_overflow_is_counted_not_written():
    metas = S.make_img_metas("tiny")
    want = _host_plan("tiny", 1, metas)
    R = want.row_query.numel()
    pl, dyn = _device_plan("tiny", 1, metas, cap=R - 7)
    c = dyn.counters.cpu().tolist()
    assert c[0] == R - 7 and c[1] == 7 and c[3] == R
    assert dyn.dropped_rows() == 7 and dyn.snapshot().dropped_rows() == 7      # (the check of the paths that never materialise)
    with pytest.raises(RuntimeError, match="row capacity"):
        dyn.materialize()


def test_plan_buffers_are_reused_and_follow_new_matrices():
    """Frame t + 1 overwrites frame t's plan in place (graph-capturable) and reflects the new rig."""
    m0 = S.make_img_metas("tiny")
    m1 = _perturbed_metas("tiny", 1, seed=5)
    pl, p0 = _device_plan("tiny", 1, m0, order="polar")
    r0 = int(p0.nrows_dev.item())
    ptr = p0.row_ref.data_ptr()
    p1 = pl.plan(m1)
    assert p1.row_ref.data_ptr() == ptr
    want = _host_plan("tiny", 1, m1)
    assert int(p1.nrows_dev.item()) == want.row_query.numel()
    assert torch.equal(p1.bev_mask.cpu(), want.bev_mask)
    assert r0 == _host_plan("tiny", 1, m0).row_query.numel()
    # device-resident matrices (the bench's graph mode) give the same plan
    m2 = [dict(lidar2img=torch.tensor(np.asarray(m1[0]["lidar2img"]), dtype=torch.float32, device=DEV),
               img_shape=m1[0]["img_shape"])]
    p2 = pl.plan(m2)
    assert int(p2.nrows_dev.item()) == want.row_query.numel()
    assert torch.equal(p2.reference_points_cam.cpu(), want.reference_points_cam)


def _overlapping_metas(name):
    """A rig whose cameras 0, 1, 2 look the same way: three cameras per visible query."""
    metas = S.make_img_metas(name)
    mats = metas[0]["lidar2img"]
    metas[0]["lidar2img"] = [mats[0], mats[0].copy(), mats[0].copy(), mats[3], mats[4], mats[5]]
    return metas


@pytest.mark.parametrize("name,temporal", [("micro4", True), ("micro", False)])
def test_encoder_on_device_plans_with_three_cameras_per_query(name, temporal):
    """fold_extra_rows: queries seen by more than two cameras (the output projection gathers two
    rows; the third is folded into the first beforehand) — against the oracle."""
    enc, sd = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=0, temporal=temporal)
    kw["img_metas"] = _overlapping_metas(name)
    want_plan = _host_plan(name, 1, kw["img_metas"])
    assert want_plan.q_rows.shape[1] >= 3
    with torch.no_grad():
        got = enc(q.to(DEV), f.to(DEV), f.to(DEV),
                  **{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}).cpu()
        want = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
    torch.testing.assert_close(got, want, rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize("name,temporal", [("micro4", True), ("tiny", True), ("micro", False)])
def test_chain_kernel_walking_every_camera_row_equals_the_fold_launch(name, temporal):
    """``modes.chain_gather_all`` (default): SpatialCrossAttention's chain kernel takes idx = q_rows_all and adds the third..
    rows of a slot itself, in the order the stand-alone fold launch adds them — bit-equal to fold + two-row gather, on a rig
    with three cameras per visible query and on the stock rig (no such slot), with and without the next layer's seam."""
    from bevformer_amd import ops
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=1, temporal=temporal)
    for metas in (_overlapping_metas(name), S.make_img_metas(name)):
        kw["img_metas"] = metas
        kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
        outs, folds = [], []
        orig = ops.fold_extra_rows
        try:
            for gather_all in (True, False):
                calls = []
                ops.fold_extra_rows = lambda *a, _c=calls, **k: (_c.append(1), orig(*a, **k))[1]
                with torch.no_grad(), ops.using(chain_gather_all=gather_all):
                    outs.append(enc(q.to(DEV), f.to(DEV), f.to(DEV), **kwd).clone())
                folds.append(len(calls))
        finally:
            ops.fold_extra_rows = orig
        assert folds[0] == 0 and folds[1] == len(enc.layers), folds       # (the launch is gone / one per layer)
        assert torch.equal(outs[0], outs[1])
        assert torch.isfinite(outs[0]).all()


@pytest.mark.parametrize("name,bs", [("micro4", 2), ("tiny", 1)])
def test_encoder_device_plans_equal_host_plans(name, bs):
    """Same frames through the torch-op plan builder (host syncs) and the device planner: the row
    order differs (image Z-order vs static polar order), the rows and therefore the output do not
    (sums of at most two camera rows commute)."""
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=2, bs=bs, temporal=True)
    kw["img_metas"] = _perturbed_metas(name, bs, seed=4)
    kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    with torch.no_grad():
        enc.device_plans = True
        a = enc(q.to(DEV), f.to(DEV), f.to(DEV), **kwd)
        enc.device_plans = False
        b = enc(q.to(DEV), f.to(DEV), f.to(DEV), **kwd)
        enc.device_plans = True
    # (the host builder evaluates torch.linspace on the GPU, the planner on the CPU: anchors differ by
    # an ulp, outputs by round-off)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


def test_device_plan_runs_without_host_sync():
    """The inference step on device plans must not synchronise with the host: it has to run under
    HIP-graph capture (a sync there raises) and the replay must follow new camera matrices."""
    enc, sd = build_pair("micro4", device=DEV)
    q, f, kw = S.make_inputs("micro4", seed=0, temporal=True)
    metas = [_perturbed_metas("micro4", 1, seed=s) for s in (0, 11)]
    l2i = torch.zeros(S.NUM_CAMS, 4, 4, device=DEV)
    kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    kwd["img_metas"] = [dict(lidar2img=l2i, img_shape=metas[0][0]["img_shape"])]
    qd, fd = q.to(DEV), f.to(DEV)

    def load(m):
        l2i.copy_(torch.tensor(np.asarray(m[0]["lidar2img"]), dtype=torch.float32))

    load(metas[0])
    with torch.no_grad():
        enc(qd, fd, fd, **kwd)                       # planner + weight caches before the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            enc(qd, fd, fd, **kwd)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = enc(qd, fd, fd, **kwd)
        for m in metas[::-1]:
            load(m)
            graph.replay()
            kw["img_metas"] = m
            want = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
            torch.testing.assert_close(out.cpu(), want, rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize("temporal", [True, False])
def test_encoder_when_no_camera_sees_anything(temporal):
    """Every pillar anchor behind every camera (depth row replaced: depth = -1 everywhere): zero ragged rows — the device plan writes a
    row count of 0, the head launches have nothing to do, the tail launches exit, the camera mean divides by the
    clamped count — and the encoder output equals the oracle's (SpatialCrossAttention contributes only its
    output-projection bias and the residual, spatial_cross_attention.py:165-173)."""
    name = "micro4"
    enc, sd = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=0, temporal=temporal)
    metas = S.make_img_metas(name)
    behind = []
    for m in metas[0]["lidar2img"]:
        m = np.array(m, dtype=np.float64)
        m[2] = [0.0, 0.0, 0.0, -1.0]          # depth -1 for every point
        behind.append(m)
    metas[0]["lidar2img"] = behind
    kw["img_metas"] = metas
    want_plan = _host_plan(name, 1, metas)
    assert want_plan.row_query.numel() == 0
    _, dyn = _device_plan(name, 1, metas)
    assert int(dyn.nrows_dev.item()) == 0
    with torch.no_grad():
        got = enc(q.to(DEV), f.to(DEV), f.to(DEV),
                  **{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}).cpu()
        want = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
    torch.testing.assert_close(got, want, rtol=5e-4, atol=5e-4)
    # and with gradients (the autograd path with an empty row list)
    for p in enc.parameters():
        p.requires_grad_(True)
    qd = q.to(DEV).requires_grad_(True)
    out = enc(qd, f.to(DEV), f.to(DEV), **{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()})
    out.sum().backward()
    assert torch.isfinite(qd.grad).all()
    torch.testing.assert_close(out.detach().cpu(), want, rtol=5e-4, atol=5e-4)


def _rank_cells(w, world, rank, layout):
    from bevformer_amd import bev_tiling
    if layout == "rows":
        h0, h1 = bev_tiling.row_blocks(w["bev_h"], world)[rank]
        return torch.arange(h0 * w["bev_w"], h1 * w["bev_w"], device=DEV)
    q0, q1 = bev_tiling.query_blocks(w["bev_h"] * w["bev_w"], world)[rank]
    return bev_tiling.sector_order(w["bev_h"], w["bev_w"], S.PC_RANGE, DEV)[1][q0:q1]


@pytest.mark.parametrize("name,world,temporal,layout", [("micro4", 2, True, "rows"), ("tiny", 5, True, "rows"), ("micro4", 3, False, "rows"),
                                                        ("tiny", 5, True, "sectors"), ("micro4", 3, True, "sectors"),
                                                        ("micro4", 2, False, "sectors")])
def test_simulated_rank_on_the_gpu_equals_its_rows_of_the_untiled_encoder(name, world, temporal, layout):
    """The N > 1 schedule on the HIP path, one rank at a time in one process (``BevTiling.simulate``): device plan
    of the rank's tile (tile-local row tables), fused kernels over the tile's rows, hoisted projections — the
    rows of the rank's shard must be the untiled encoder's rows.  (With history only: without it TemporalSelfAttention
    samples the CURRENT grid, which a simulated rank does not have beyond layer 0 — there the first layer's rows
    are what can be compared, so that case runs a 1-layer encoder.)"""
    from bevformer_amd import bev_tiling
    enc, _ = build_pair(name, device=DEV)
    if not temporal:
        enc.layers = enc.layers[:1]
        enc.num_layers = 1
    q, f, kw = S.make_inputs(name, seed=0, temporal=temporal, device=DEV)
    w = S.WORKLOADS[name]
    with torch.no_grad():
        want = enc(q, f, f, **kw)
        for rank in range(world):
            bev_tiling.enable_bev_tiling(enc, simulate=(rank, world), layout=layout)
            got = enc(q, f, f, **kw)
            bev_tiling.disable_bev_tiling(enc)
            mine = _rank_cells(w, world, rank, layout)
            torch.testing.assert_close(got[:, mine], want[:, mine], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name,world,layout", [("tiny", 5, "rows"), ("micro4", 3, "rows"), ("tiny", 5, "sectors"), ("tiny", 8, "sectors")])
def test_tiled_rank_skips_the_value_projection_of_cameras_it_cannot_see(name, world, layout):
    """BEV tiling over GPUs: the camera-feature value projection is replicated work, but a rank's tile only samples
    cameras that some of its queries project into — the tile plan's device-side camera starts gate the projection's
    workgroups (``ops.linear(segments=...)``).  With the projection's output pre-filled with NaN, every rank's rows
    must still be the untiled encoder's rows (nobody reads a skipped camera), and across the ranks some camera must
    actually have been skipped (else this test shows nothing)."""
    from bevformer_amd import bev_tiling, ops
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=0, temporal=True, device=DEV)
    w = S.WORKLOADS[name]
    seen, real = [], ops.linear

    def spy(*a, **k):
        if k.get("segments") is not None:
            st = k["segments"][0]
            seen.append(int(((st[1:] - st[:-1]) == 0).sum()))
        return real(*a, **k)

    ops._SEGMENT_POISON.update(on=True, launches=0)
    ops.linear = spy
    try:
        with torch.no_grad(), ops.using(gemm_kernel="panel64"):
            want = enc(q, f, f, **kw)
            assert not seen                         # the untiled encoder projects every camera
            for rank in range(world):
                bev_tiling.enable_bev_tiling(enc, simulate=(rank, world), layout=layout)
                got = enc(q, f, f, **kw)
                bev_tiling.disable_bev_tiling(enc)
                mine = _rank_cells(w, world, rank, layout)
                assert torch.isfinite(got[:, mine]).all()
                torch.testing.assert_close(got[:, mine], want[:, mine], rtol=1e-4, atol=1e-4)
    finally:
        ops.linear = real
        ops._SEGMENT_POISON["on"] = False
    assert len(seen) == world and ops._SEGMENT_POISON["launches"] == world
    assert sum(seen) > 0, f"no rank of {world} could skip a camera on workload {name}: {seen}"


@pytest.mark.parametrize("bs", [1, 2])
def test_hybrid_anchors_from_one_launch_equal_the_reference_statement(bs):
    """``geometry.hybrid_ref_2d``: stack([ref_2d + shift, ref_2d], 1) (encoder.py:226-237) written by one elementwise launch in
    the sampling kernels' row layout — bit-equal values in the reference's layout, and (bs = 1) TemporalSelfAttention's row
    layout of it is the same memory, not a copy."""
    from bevformer_amd.modules.temporal_self_attention import _rows_layout
    Q = 50 * 50
    g = torch.Generator().manual_seed(5)
    ref = torch.rand(bs, Q, 1, 2, generator=g).to(DEV)
    shift = (torch.randn(bs, 2, generator=g) * 0.05).to(DEV)
    want = torch.stack([ref + shift[:, None, None, :], ref], 1).reshape(bs * 2, Q, 1, 2)
    got = G.hybrid_ref_2d(ref, shift)
    assert got.shape == want.shape and torch.equal(got, want)
    rows = _rows_layout(got, bs, 2, Q, 1)
    assert rows.is_contiguous() and torch.equal(rows, _rows_layout(want, bs, 2, Q, 1))
    if bs == 1:
        assert rows.data_ptr() == got.data_ptr()
    first_want = torch.stack([ref, ref], 1).reshape(bs * 2, Q, 1, 2)           # (a scene's first frame)
    first = G.hybrid_ref_2d(ref, None)
    assert first.shape == first_want.shape and torch.equal(first, first_want)
    assert torch.equal(_rows_layout(first, bs, 2, Q, 1), _rows_layout(first_want, bs, 2, Q, 1))


@pytest.mark.parametrize("name,temporal", [("small4", True), ("micro4", False)])
def test_plan_kernels_on_the_side_stream_equal_the_main_stream_schedule(name, temporal):
    """``modes.plan_on_side`` (default with ``overlap_value_proj``): the frame-plan kernels run on the side stream ahead of the
    hoisted camera-value projection and meet the main stream at the first SpatialCrossAttention.  Eight frames back to back,
    NEW camera matrices in the same device tensor before every frame and no synchronisation in between — a plan that overwrote
    its buffers under the previous frame's readers, or a reader that ran ahead of its plan, would show — against the
    one-stream schedule, frame by frame, bit for bit."""
    from bevformer_amd import ops
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=4, temporal=temporal)
    rigs = [torch.tensor(np.asarray(_perturbed_metas(name, 1, seed=s)[0]["lidar2img"]), dtype=torch.float32, device=DEV)
            for s in range(8)]
    l2i = torch.zeros_like(rigs[0])
    kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    kwd["img_metas"] = [dict(lidar2img=l2i, img_shape=S.make_img_metas(name)[0]["img_shape"])]
    qd, fd = q.to(DEV), f.to(DEV)
    outs = {}
    for side in (False, True):
        with torch.no_grad(), ops.using(overlap_value_proj=True, plan_on_side=side):
            l2i.copy_(rigs[0])
            enc(qd, fd, fd, **kwd)                  # (planner, weight images)
            torch.cuda.synchronize()
            frames = []
            for r in rigs:
                l2i.copy_(r)
                frames.append(enc(qd, fd, fd, **kwd).clone())
            torch.cuda.synchronize()
        outs[side] = frames
    for i in range(len(rigs)):
        assert torch.equal(outs[True][i], outs[False][i]), i
    assert not torch.equal(outs[True][0], outs[True][1])        # (the rigs do differ)


@pytest.mark.parametrize("name", ["tiny", "small4"])
def test_capacity_sized_sampling_launch_equals_hint_plus_tail(name):
    """``modes.fused_capacity_launch``: SpatialCrossAttention's sampling over the device-side row count as ONE launch sized by
    the row capacity (the "auto" default where the surplus is small: tiny; not at small4), as a hint-sized launch + a strided
    tail, and by the default policy — the same rows, bit for bit."""
    from bevformer_amd import ops
    from bevformer_amd.ops import sampling
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=6, temporal=True)
    kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    outs = []
    for mode in (False, True, "auto"):
        with torch.no_grad(), ops.using(fused_capacity_launch=mode):
            outs.append(enc(q.to(DEV), f.to(DEV), f.to(DEV), **kwd).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    (planner,) = enc._planners.values()
    surplus = planner.cap - planner.launch_rows
    assert (surplus <= sampling.FUSED_CAPACITY_AUTO_ROWS) == (name == "tiny"), surplus
