"""GPU, ONE device: the N > 1 path of BASELINE configs[3] / [4] executed for real on a 1-GPU box.

Two (three) processes share ``cuda:0``; RCCL refuses duplicate devices, so the exchange step runs over ``gloo`` with the
shards staged through host memory (``bev_tiling.all_gather_rows``: a functional path, never a measured one).
Everything else is the product path — per-rank device-side tile plans, camera skipping, the HIP kernels, the sector
gather / scatter — so this is the test that the tiled schedule, as a whole, reproduces the single-process encoder ON
THE GPU, and that the history queue (``BevHistory`` -> ``get_bev_features``) over the tiled encoder reproduces the
untiled queue.  The RCCL transport itself is tests/test_tiling_rccl_gpu.py (needs >= 2 devices).

Second half: ``GraphedBevHistory`` (the step replayed from HIP graphs) over a SIMULATED rank of a tiled job in one
process: with the full history written back between frames, the rank's own rows of every frame must be the untiled
queue's rows."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch.device("cuda", 0)


def _encoder_worker(rank, world, port, name, temporal, layout, ret):
    dev = _init(rank, world, port)
    try:
        from helpers import build_pair
        from bevformer_amd import bev_tiling
        from bevformer_amd import synthetic as S
        enc, _ = build_pair(name, device=dev)
        q, f, kw = S.make_inputs(name, seed=0, temporal=temporal, device=dev)
        with torch.no_grad():
            want = enc(q, f, f, **kw)
            bev_tiling.enable_bev_tiling(enc, layout=layout)
            got = enc(q, f, f, **kw)
            again = enc(q, f, f, **kw)
            bev_tiling.disable_bev_tiling(enc)
        host = got.cpu()
        gathered = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(gathered, host)
        ret[rank] = ((got - want).abs().max().item(), all(torch.equal(g, gathered[0]) for g in gathered),
                     bool(torch.equal(got, again)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,temporal,layout,world", [("micro4", True, "rows", 2), ("micro4", False, "sectors", 2),
                                                        ("tiny", True, "sectors", 3),
                                                        # BASELINE configs[3]'s size: 200 x 200 queries, 4 levels, 6 layers
                                                        ("base", True, "rows", 2)])
def test_tiled_encoder_on_ranks_sharing_one_gpu(name, temporal, layout, world):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_encoder_worker, args=(world, _free_port(), name, temporal, layout, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err, same, stable = ret[r]
        assert same, "ranks disagree on the gathered grid"
        assert stable, "the tiled step is not reproducible"
        assert err < 5e-4, f"rank {r}: tiled vs untiled max abs {err:.2e}"


def _queue_worker(rank, world, port, name, layout, ret):
    dev = _init(rank, world, port)
    try:
        from helpers import build_transformer_pair
        from test_history_cpu import _video
        from bevformer_amd import bev_tiling, history
        t, _ = build_transformer_pair(name, device=dev)
        frames = _video(name, 4, scene_break=2)

        def run():
            hist = history.BevHistory()
            outs = []
            for mlvl, metas, bq, kw in frames:
                def fn(f, m, p, bq=bq, kw=kw):
                    return t.get_bev_features([x.to(dev) for x in f], bq.to(dev), kw["bev_h"], kw["bev_w"],
                                              grid_length=kw["grid_length"], bev_pos=kw["bev_pos"].to(dev), prev_bev=p,
                                              img_metas=m)
                outs.append(hist.step(fn, mlvl, metas).clone())
            return outs
        want = run()
        bev_tiling.enable_bev_tiling(t.encoder, layout=layout)
        got = run()
        bev_tiling.disable_bev_tiling(t.encoder)
        ret[rank] = [(g - w).abs().max().item() for g, w in zip(got, want)]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["rows", "sectors"])
def test_history_queue_over_the_tiled_encoder_on_ranks_sharing_one_gpu(layout):
    """BASELINE configs[4] (4-frame history queue, BEV-tiled encoder) on the GPU kernels, 2 ranks: frames 0 and 2 open a
    scene (per-layer exchange), frames 1 and 3 sample the rotated, all-gathered BEV of the frame before."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_queue_worker, args=(world, _free_port(), "micro4", layout, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert max(ret[r]) < 1e-3, (r, ret[r])


@pytest.mark.parametrize("layout,world", [("rows", 2), ("sectors", 4)])
def test_graphed_history_queue_over_a_simulated_rank(layout, world):
    """``GraphedBevHistory`` over rank r of a ``world``-rank tiled job simulated in this process (``BevTiling.simulate``:
    the all-gather is the copy of the rank's own shard).  The history a frame reads is the FULL previous BEV, which a
    lone simulated rank cannot produce — the test writes the untiled queue's BEV into the graphed queue's history
    buffer between frames; then the rank's own rows of every frame WITH history (replayed from its HIP graphs, tile plan
    and camera skipping included) must equal the untiled queue's rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import build_transformer_pair
    from test_history_cpu import _video
    from bevformer_amd import bev_tiling, history
    from bevformer_amd import synthetic as S
    dev = torch.device("cuda:0")
    name = "micro4"
    w = S.WORKLOADS[name]
    frames = _video(name, 5, scene_break=3)
    t, _ = build_transformer_pair(name, device=dev)
    mlvl0, _, bq, kw = frames[0]
    feats = [x.to(dev) for x in mlvl0]
    bq_d, pos_d = bq.to(dev), kw["bev_pos"].to(dev)

    def bev_fn(f, m, p):
        return t.get_bev_features(f, bq_d, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"], bev_pos=pos_d,
                                  prev_bev=p, img_metas=m)

    eager = history.BevHistory()
    want = [eager.step(bev_fn, [x.to(dev) for x in mlvl], metas).clone() for mlvl, metas, _, _ in frames]
    for rank in (0, world - 1):
        bev_tiling.enable_bev_tiling(t.encoder, simulate=(rank, world), layout=layout)
        graphed = history.GraphedBevHistory(bev_fn, feats)
        if layout == "rows":
            h0, h1 = bev_tiling.row_blocks(w["bev_h"], world)[rank]
            mine = torch.arange(h0 * w["bev_w"], h1 * w["bev_w"], device=dev)
        else:
            q0, q1 = bev_tiling.query_blocks(w["bev_h"] * w["bev_w"], world)[rank]
            mine = bev_tiling.sector_order(w["bev_h"], w["bev_w"], S.PC_RANGE, dev)[1][q0:q1]
        for i, (mlvl, metas, _, _) in enumerate(frames):
            got = graphed.step(None, [x.to(dev) for x in mlvl], metas).clone()
            if i not in (0, 3):
                # (a frame that opens a scene exchanges the CURRENT grid after every layer — a lone simulated rank sees only
                # its own shard there, so only frames with a history BEV are comparable; the first-frame graph still runs)
                torch.testing.assert_close(got[:, mine], want[i][:, mine], rtol=1e-4, atol=1e-4)
            graphed.prev.copy_(want[i])         # the all-gathered BEV the next frame's history would be
        bev_tiling.disable_bev_tiling(t.encoder)
