"""BEV-query tiling over world_size 2 (gloo, CPU): the tiled schedule must
reproduce the single-process encoder.  The operator calls are
routed through the CPU oracle (tests/helpers.oracle_ops) because the product
operator has no CPU path; what is under test here is the host logic of the
N > 1 path: row blocks, plan slicing, per-layer exchange without history, and
the final all-gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, temporal, bs, layout, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import build_pair, oracle_ops
        from bevformer_amd import bev_tiling
        from bevformer_amd import synthetic as S
        enc, _ = build_pair(name)
        q, f, kw = S.make_inputs(name, seed=0, temporal=temporal, bs=bs)
        with oracle_ops(), torch.no_grad():
            want = enc(q, f, f, **kw)
            bev_tiling.enable_bev_tiling(enc, layout=layout)
            got = enc(q, f, f, **kw)
            bev_tiling.disable_bev_tiling(enc)
        err = (got - want).abs().max().item()
        # every rank must hold the identical full grid
        gathered = [torch.empty_like(got) for _ in range(world)]
        dist.all_gather(gathered, got)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        ret[rank] = (err, same, tuple(got.shape))
    finally:
        dist.destroy_process_group()


def _queue_worker(rank, world, port, name, layout, ret):
    """BASELINE configs[4] as one thing: the test-time history queue (``BevHistory.step`` = detectors/bevformer.py:236-269)
    driving ``PerceptionTransformer.get_bev_features`` (modules/transformer.py:104-200) over the BEV-TILED encoder."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import build_transformer_pair, oracle_ops
        from test_history_cpu import _video
        from bevformer_amd import bev_tiling, history
        t, _ = build_transformer_pair(name)
        frames = _video(name, 4, scene_break=2)

        def run():
            hist = history.BevHistory()
            outs, seen_prev = [], []
            for mlvl, metas, bq, kw in frames:
                def fn(f, m, p, bq=bq, kw=kw):
                    seen_prev.append(p is not None)
                    return t.get_bev_features(f, bq, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"],
                                              bev_pos=kw["bev_pos"], prev_bev=p, img_metas=m)
                outs.append(hist.step(fn, mlvl, metas).clone())
            return outs, seen_prev
        with oracle_ops(), torch.no_grad():
            want, seen_w = run()
            bev_tiling.enable_bev_tiling(t.encoder, layout=layout)
            got, seen_g = run()
            bev_tiling.disable_bev_tiling(t.encoder)
        assert seen_w == seen_g == [False, True, False, True]       # frames 0 and 2 open a scene: no history
        errs = [(g - w).abs().max().item() for g, w in zip(got, want)]
        last = got[-1].contiguous()
        gathered = [torch.empty_like(last) for _ in range(world)]
        dist.all_gather(gathered, last)
        ret[rank] = (errs, all(torch.equal(g, gathered[0]) for g in gathered))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layout", [(2, "rows"), (2, "sectors"), (3, "sectors")])
def test_history_queue_over_the_tiled_encoder_matches_the_untiled_queue(world, layout):
    """configs[4]: four frames with a scene break through ``BevHistory`` -> ``get_bev_features`` -> the tiled encoder on
    ``world`` gloo ranks == the same queue over the untiled encoder; every frame's BEV (the next frame's history, after
    the caller's rotation / shift) is the all-gathered full grid on every rank, and frames without history take the
    per-layer exchange."""
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_queue_worker, args=(world, port, "micro", layout, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        errs, same = ret[r]
        assert same, "ranks disagree on the last frame's BEV"
        assert max(errs) < 5e-5, (r, errs)


@pytest.mark.parametrize("name,temporal,bs,world,layout", [("micro", True, 1, 2, "rows"), ("micro", False, 1, 2, "rows"),
                                                           ("micro4", True, 2, 2, "rows"), ("micro", True, 1, 5, "rows"),
                                                           ("micro", True, 1, 2, "sectors"), ("micro", False, 1, 3, "sectors"),
                                                           ("micro4", True, 2, 5, "sectors")])
def test_two_rank_tiling_matches_single(name, temporal, bs, world, layout):
    """world 2 (even row blocks) and world 5 (12 BEV rows -> blocks of 3, 3, 2, 2, 2: padded shards
    in the all-gather); the sector layout (queries in azimuth order, output back in grid order) with and without
    history and with uneven shards."""
    from bevformer_amd import synthetic as S
    try:
        S.make_inputs(name, seed=0, temporal=temporal, bs=bs)
    except TypeError:
        pytest.skip("synthetic.make_inputs has no bs argument")
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, name, temporal, bs, layout, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err, same, shape = ret[r]
        assert same, "ranks disagree on the gathered BEV grid"
        # row-wise ops only: tiled == untiled up to GEMM blocking round-off (fp32)
        assert err < 2e-5, (r, err)


def test_row_blocks_cover_grid_unevenly():
    from bevformer_amd.bev_tiling import row_blocks
    for h in (1, 7, 50, 200):
        for w in (1, 2, 3, 8):
            b = row_blocks(h, w)
            assert len(b) == w and b[0][0] == 0 and b[-1][1] == h
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [x1 - x0 for x0, x1 in b]
            assert max(sizes) - min(sizes) <= 1


def test_sector_permutation_is_a_permutation_by_azimuth():
    sys.path.insert(0, ROOT)
    import numpy as np
    from bevformer_amd.modules.geometry import sector_permutation
    from bevformer_amd import bev_tiling
    pc = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    for h, w in ((12, 10), (50, 50), (7, 9)):
        perm = sector_permutation(h, w, pc)
        assert sorted(perm.tolist()) == list(range(h * w))
        xs = (np.arange(w) + 0.5) / w * 102.4 - 51.2
        ys = (np.arange(h) + 0.5) / h * 102.4 - 51.2
        az = np.arctan2(np.repeat(ys, w), np.tile(xs, h))[perm.numpy()]
        assert (np.diff(az) >= -1e-8).all()
        for world in (2, 3, 8):
            b = bev_tiling.query_blocks(h * w, world)
            assert b[0][0] == 0 and b[-1][1] == h * w and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(q1 - q0 for q0, q1 in b) - min(q1 - q0 for q0, q1 in b) <= 1


@pytest.mark.parametrize("world,layout", [(2, "rows"), (5, "rows"), (3, "sectors")])
def test_simulated_rank_equals_its_rows_of_the_untiled_encoder(world, layout):
    """``BevTiling.simulate = (rank, world)`` (bench.py's ``multi_gpu_model`` / ``--simulate-rank``): one process, no
    process group, the all-gather replaced by the copy of the rank's own shard — the rows of that shard must be the
    untiled encoder's rows, for every rank (host logic on the CPU, operator calls through the oracle)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import build_pair, oracle_ops
    from bevformer_amd import bev_tiling
    from bevformer_amd import synthetic as S
    name = "micro"
    enc, _ = build_pair(name)
    q, f, kw = S.make_inputs(name, seed=0, temporal=True)
    w = S.WORKLOADS[name]
    with oracle_ops(), torch.no_grad():
        want = enc(q, f, f, **kw)
        for rank in range(world):
            bev_tiling.enable_bev_tiling(enc, simulate=(rank, world), layout=layout)
            got = enc(q, f, f, **kw)
            bev_tiling.disable_bev_tiling(enc)
            if layout == "rows":
                h0, h1 = bev_tiling.row_blocks(w["bev_h"], world)[rank]
                mine = torch.arange(h0 * w["bev_w"], h1 * w["bev_w"])
            else:
                q0, q1 = bev_tiling.query_blocks(w["bev_h"] * w["bev_w"], world)[rank]
                mine = bev_tiling.sector_order(w["bev_h"], w["bev_w"], S.PC_RANGE, "cpu")[1][q0:q1]
            other = torch.ones(got.shape[1], dtype=torch.bool)
            other[mine] = False
            torch.testing.assert_close(got[:, mine], want[:, mine], rtol=1e-5, atol=1e-5)
            assert (got[:, other] == 0).all()
