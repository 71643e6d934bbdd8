"""GPU: a short video through the test-time history queue (``BevHistory`` around the product
``PerceptionTransformer.get_bev_features`` on the HIP kernels) against the oracle's restatement of
detectors/bevformer.py:236-269 around the oracle's ``get_bev_features``."""
import pytest
import torch

from bevformer_amd import history
from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O

from helpers import build_transformer_pair, kernel_rotation_index, split_transformer_sd
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
    ties, pixels = [], []
    for mlvl, metas, bq, kw in frames:
        def product(f, m, p):
            return t.get_bev_features([x.to(DEV) for x in f], bq.to(DEV), kw["bev_h"], kw["bev_w"],
                                      grid_length=kw["grid_length"], bev_pos=kw["bev_pos"].to(DEV), prev_bev=p,
                                      img_metas=m)

        def rotate_as_the_kernel(img, angle, center):
            # Nearest-neighbour rotation is an index map, and a coordinate within fp32 round-off of a rounding tie
            # may pick the other neighbour on the GPU (no fused multiply-adds in the kernel, bmm on the CPU): the
            # oracle frame takes the KERNEL's map, the pixels where it differs from the oracle's own map are counted
            # as ties (each must be an adjacent source pixel: tests/test_prologue_gpu.py), and then EVERY row of
            # the frame has to agree.
            C, h, ww = img.shape
            idx = kernel_rotation_index(h, ww, angle, center, DEV)
            ties.append(int((idx != O.rotate_source_index(h, ww, angle, list(center))).sum()))
            pixels.append(h * ww)
            flat = img.reshape(C, h * ww)
            return (flat[:, idx.clamp(min=0)] * (idx >= 0).to(img.dtype)).view(C, h, ww)

        def oracle(f, m, p):
            return O.get_bev_features(own, enc, f, bq, kw["bev_h"], kw["bev_w"], bev_pos=kw["bev_pos"], img_metas=m,
                                      pc_range=S.PC_RANGE, grid_length=kw["grid_length"], prev_bev=p,
                                      rotate_center=(w["bev_w"] // 2, w["bev_h"] // 2), rotate_fn=rotate_as_the_kernel)
        import copy
        got = hist.step(product, mlvl, metas).cpu()
        with torch.no_grad():
            want = O.forward_test_step(info, oracle, mlvl, copy.deepcopy(metas))
        torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-3)
    print(f"{name}: rotation ties over the video: {sum(ties)} of {sum(pixels)} history pixels")
    assert sum(ties) <= max(1, int(2e-4 * sum(pixels)))


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
        # device float64 cos / sin against the host's: the two rotation matrices may differ in the last bit and move a
        # history pixel that sits on a rounding tie.  Count those pixels from the two index maps; without any, every
        # row must agree, with some, only rows in reach of a moved pixel may differ.
        err = (got - want).abs().amax(-1)
        angle = float(eager.rewritten_metas[0]["can_bus"][-1])
        ctr = (kw["bev_w"] // 2, kw["bev_h"] // 2)
        moved = int((kernel_rotation_index(kw["bev_h"], kw["bev_w"], angle, ctr, DEV)
                     != kernel_rotation_index(kw["bev_h"], kw["bev_w"], angle, ctr, DEV, device_pose=True)).sum())
        assert int((err > 1e-3).sum()) <= 512 * moved, (moved, err.max().item())
    assert set(graphed.graphs) == {False, True}


def test_graphed_queue_forms_agree_and_the_staging_ring_wraps():
    """The captured frames as one-stream graphs (the queue's default) and as two-stream graphs
    (``overlap_value_proj=True``), each fed through the ring of pinned pose slots for MORE frames than the ring
    has slots (a slot is rewritten only after the copy that read it ran), and the blocking-copy form
    (``STAGING_SLOTS = 0``): the same BEV, bit for bit, frame by frame."""
    name = "micro4"
    frames = _video(name, 2 * history.GraphedBevHistory.STAGING_SLOTS + 3, scene_break=9)
    t, _ = build_transformer_pair(name, device=DEV)
    mlvl0, _, bq, kw = frames[0]
    feats = [x.to(DEV) for x in mlvl0]
    bq_d, pos_d = bq.to(DEV), kw["bev_pos"].to(DEV)

    def bev_fn(f, m, p):
        return t.get_bev_features(f, bq_d, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"],
                                  bev_pos=pos_d, prev_bev=p, img_metas=m)

    one = history.GraphedBevHistory(bev_fn, feats)
    two = history.GraphedBevHistory(bev_fn, feats, overlap_value_proj=True)
    blocking = history.GraphedBevHistory(bev_fn, feats)
    blocking.STAGING_SLOTS = 0
    assert one.overlap_value_proj is False and two.overlap_value_proj is True
    outs = {k: [] for k in ("one", "two", "blocking")}
    for mlvl, metas, _, _ in frames:           # no synchronisation between frames: the host runs ahead through the ring
        f = [x.to(DEV) for x in mlvl]
        outs["one"].append(one.step(None, f, metas).clone())
    for mlvl, metas, _, _ in frames:
        outs["two"].append(two.step(None, [x.to(DEV) for x in mlvl], metas).clone())
    for mlvl, metas, _, _ in frames:
        outs["blocking"].append(blocking.step(None, [x.to(DEV) for x in mlvl], metas).clone())
    torch.cuda.synchronize()
    assert one._staged > len(one._staging) > 0 and not blocking._staging
    for i in range(len(frames)):
        assert torch.equal(outs["one"][i], outs["blocking"][i]), i
        assert torch.equal(outs["one"][i], outs["two"][i]), i
