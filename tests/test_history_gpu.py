"""GPU: a short video through the test-time history queue (``BevHistory`` around the product
``PerceptionTransformer.get_bev_features`` on the HIP kernels) against the oracle's restatement of
detectors/bevformer.py:236-269 around the oracle's ``get_bev_features``."""
import pytest
import torch

from bevformer_amd import history
from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O

from helpers import build_transformer_pair, split_transformer_sd
from test_history_cpu import _video

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("name", ["micro4", "tiny"])
def test_video_through_the_history_queue(name):
    frames = _video(name, 4, scene_break=3)
    t, sd = build_transformer_pair(name, device=DEV)
    own, enc = split_transformer_sd(sd)
    w = S.WORKLOADS[name]
    hist = history.BevHistory()
    info = {"prev_bev": None, "scene_token": None, "prev_pos": 0, "prev_angle": 0}
    for mlvl, metas, bq, kw in frames:
        def product(f, m, p):
            return t.get_bev_features([x.to(DEV) for x in f], bq.to(DEV), kw["bev_h"], kw["bev_w"],
                                      grid_length=kw["grid_length"], bev_pos=kw["bev_pos"].to(DEV), prev_bev=p,
                                      img_metas=m)

        def oracle(f, m, p):
            return O.get_bev_features(own, enc, f, bq, kw["bev_h"], kw["bev_w"], bev_pos=kw["bev_pos"], img_metas=m,
                                      pc_range=S.PC_RANGE, grid_length=kw["grid_length"], prev_bev=p,
                                      rotate_center=(w["bev_w"] // 2, w["bev_h"] // 2))
        import copy
        got = hist.step(product, mlvl, metas).cpu()
        with torch.no_grad():
            want = O.forward_test_step(info, oracle, mlvl, copy.deepcopy(metas))
        # a rotation tie may move single history rows by one pixel (tests/test_prologue_gpu.py): bound
        # the fraction of rows out of tolerance instead of every element
        err = (got - want).abs().amax(-1)
        assert (err > 1e-3).float().mean().item() < 5e-3, err.max().item()


@pytest.mark.parametrize("name", ["micro4", "tiny"])
def test_graph_replayed_history_queue_equals_the_eager_one(name):
    """``GraphedBevHistory``: prologue (pose from DEVICE tensors: shift, rotation matrix, can-bus MLP input),
    frame plan and encoder replayed from two captured HIP graphs, against ``BevHistory`` launching the same
    modules eagerly with the pose read on the host — a video of 6 frames with a scene break (both graphs are
    captured and then replayed with new poses and camera matrices)."""
    frames = _video(name, 6, scene_break=4)
    t, _ = build_transformer_pair(name, device=DEV)
    mlvl0, _, bq, kw = frames[0]
    feats = [x.to(DEV) for x in mlvl0]
    bq_d, pos_d = bq.to(DEV), kw["bev_pos"].to(DEV)       # (no host -> device copies inside a captured step)

    def bev_fn(f, m, p):
        return t.get_bev_features(f, bq_d, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"],
                                  bev_pos=pos_d, prev_bev=p, img_metas=m)

    eager = history.BevHistory()
    graphed = history.GraphedBevHistory(bev_fn, feats)
    for mlvl, metas, _, _ in frames:
        f = [x.to(DEV) for x in mlvl]
        want = eager.step(bev_fn, f, metas).clone()
        got = graphed.step(None, f, metas).clone()
        # (device float64 cos / sin against the host's: a rotation tie may move single history rows)
        err = (got - want).abs().amax(-1)
        assert (err > 1e-3).float().mean().item() < 5e-3, err.max().item()
    assert set(graphed.graphs) == {False, True}
