"""GPU, two or more devices: the BEV-tiled schedule over RCCL (``torch.distributed`` backend ``nccl`` = RCCL on ROCm,
one process per GPU) — the real ``all_gather_into_tensor`` over xGMI — must reproduce the single-GPU encoder, and every
rank must hold the identical grid.  Skipped where fewer than two GPUs are visible (a ``gpurun`` box has one; the host
logic of the N > 1 path is covered on CPU by tests/test_tiling_gloo.py).  SURVEY.md §8e."""
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


def _worker(rank, world, port, name, temporal, layout, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
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
            bev_tiling.disable_bev_tiling(enc)
        err = (got - want).abs().max().item()
        gathered = [torch.empty_like(got) for _ in range(world)]
        dist.all_gather(gathered, got)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        ret[rank] = (err, same, tuple(got.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,temporal,layout", [("tiny", True, "rows"), ("tiny", False, "rows"), ("micro4", True, "rows"),
                                                  ("tiny", True, "sectors"), ("tiny", False, "sectors")])
def test_tiled_encoder_over_rccl_matches_single_gpu(name, temporal, layout):
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs at least two GPUs (RCCL all-gather between processes)")
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), name, temporal, layout, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        err, same, _ = ret[rank]
        assert same, "ranks disagree on the gathered grid"
        assert err < 5e-4, f"rank {rank}: tiled vs untiled max abs {err:.2e}"
