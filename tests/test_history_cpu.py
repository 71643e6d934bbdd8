"""History-BEV queue: the product's two state machines
(``bevformer_amd.history``) against the oracle's restatement of
detectors/bevformer.py:158-177 / :236-269, with the encoder's operator calls routed through the
CPU oracle (the product operator has no CPU path)."""
import copy

import numpy as np
import torch

from bevformer_amd import history
from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O

from helpers import build_transformer_pair, oracle_ops, split_transformer_sd


def _video(name, n_frames, scene_break=None):
    """Per-frame (features, metas) of one stream with ABSOLUTE can-bus poses."""
    frames = []
    pos = np.zeros(3)
    yaw = 10.0
    for t in range(n_frames):
        mlvl, bq, kw = S.make_transformer_inputs(name, seed=10 + t, bs=1, temporal=False)
        meta = kw["img_metas"][0]
        pos = pos + np.array([1.5 + 0.1 * t, 0.4, 0.0])
        yaw = yaw + 3.0
        meta["can_bus"][:3] = pos
        meta["can_bus"][-1] = yaw
        meta["scene_token"] = "scene-b" if scene_break is not None and t >= scene_break else "scene-a"
        meta["prev_bev_exists"] = not (t == 0 or t == scene_break)
        frames.append((mlvl, [meta], bq, kw))
    return frames


def _bev_fns(name):
    t, sd = build_transformer_pair(name)
    own, enc = split_transformer_sd(sd)
    w = S.WORKLOADS[name]

    def product(mlvl, metas, prev_bev, bq, kw):
        return t.get_bev_features(mlvl, bq, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"],
                                  bev_pos=kw["bev_pos"], prev_bev=prev_bev, img_metas=metas)

    def oracle(mlvl, metas, prev_bev, bq, kw):
        return O.get_bev_features(own, enc, mlvl, bq, kw["bev_h"], kw["bev_w"], bev_pos=kw["bev_pos"],
                                  img_metas=metas, pc_range=S.PC_RANGE, grid_length=kw["grid_length"],
                                  prev_bev=prev_bev, rotate_center=(w["bev_w"] // 2, w["bev_h"] // 2))
    return product, oracle


def test_test_time_history_follows_the_reference_state_machine():
    name = "micro"
    frames = _video(name, 4, scene_break=2)
    product, oracle = _bev_fns(name)
    hist = history.BevHistory()
    info = {"prev_bev": None, "scene_token": None, "prev_pos": 0, "prev_angle": 0}
    with oracle_ops(), torch.no_grad():
        for t, (mlvl, metas, bq, kw) in enumerate(frames):
            want_metas = copy.deepcopy(metas)
            got = hist.step(lambda f, m, p: product(f, m, p, bq, kw), mlvl, metas)
            want = O.forward_test_step(info, lambda f, m, p: oracle(f, m, p, bq, kw), mlvl, want_metas)
            # the deltas handed to the encoder's caller are the reference's
            np.testing.assert_array_equal(hist.rewritten_metas[0]["can_bus"], want_metas[0]["can_bus"])
            if t in (0, 2):
                assert (hist.rewritten_metas[0]["can_bus"][:3] == 0).all()       # first frame of a scene
            torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
            # the caller's own metas are left alone (the reference rewrites them in place)
            assert metas[0]["can_bus"][0] != hist.rewritten_metas[0]["can_bus"][0] or t in ()
    assert hist.prev_frame_info["scene_token"] == "scene-b"
    np.testing.assert_array_equal(hist.prev_frame_info["prev_pos"], info["prev_pos"])


def test_video_test_mode_off_never_uses_history():
    name = "micro"
    frames = _video(name, 2)
    product, _ = _bev_fns(name)
    seen = []
    hist = history.BevHistory(video_test_mode=False)
    with oracle_ops(), torch.no_grad():
        for mlvl, metas, bq, kw in frames:
            hist.step(lambda f, m, p: (seen.append(p), product(f, m, p, bq, kw))[1], mlvl, metas)
    assert seen == [None, None]


def test_training_queue_history_matches_the_reference_loop():
    name = "micro"
    frames = _video(name, 3, scene_break=1)
    product, oracle = _bev_fns(name)
    # queue tensors: per level (bs, len_queue, Nc, C, h, w); metas: per batch element {i: meta}
    feats_queue = [torch.stack([f[0][lvl] for f in frames], 1) for lvl in range(len(frames[0][0]))]
    metas_list = [{i: f[1][0] for i, f in enumerate(frames)}]
    # every frame of the queue shares the query / positional tensors of frame 0 (one head)
    bq, kw = frames[0][2], frames[0][3]
    with oracle_ops():
        got = history.obtain_history_bev(lambda f, m, p: product(f, m, p, bq, kw), feats_queue, metas_list)
        want = O.obtain_history_bev(lambda f, m, p: oracle(f, m, p, bq, kw), feats_queue, copy.deepcopy(metas_list))
    assert not got.requires_grad
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
