"""The history-BEV queue around the encoder.

The reference keeps this logic in its detector (projects/mmdet3d_plugin/bevformer/detectors/
bevformer.py): ``obtain_history_bev`` (:158-177) walks the earlier frames of a training queue
under ``no_grad`` and hands the last BEV to the differentiable frame; ``forward_test`` (:236-269)
carries ``prev_frame_info`` between calls — scene reset, and the can-bus pose turned into a delta
against the previous frame.  Both reach the encoder through ``pts_bbox_head(..., only_bev=True)``
-> ``PerceptionTransformer.get_bev_features``.  Here the same two state machines drive any
callable ``bev_fn(mlvl_feats, img_metas, prev_bev) -> (bs, Q, C)`` (e.g. a partial of
``PerceptionTransformer.get_bev_features``); backbone, head and boxes stay out of scope.
"""
import copy

import torch


def obtain_history_bev(bev_fn, feats_queue, img_metas_list):
    """detectors/bevformer.py:158-177.

    ``feats_queue``: per level a tensor (bs, len_queue, Nc, C, h, w) — the backbone features of
    the queue's earlier frames (the reference slices ``each_scale[:, i]``, :173);
    ``img_metas_list``: per batch element a list / dict of per-frame metas (``each[i]``, :170).
    Frames are walked in order without gradients; a frame whose meta says
    ``prev_bev_exists == False`` starts a new scene (:171-172).  Returns the last BEV or None
    for an empty queue."""
    prev_bev = None
    len_queue = feats_queue[0].shape[1] if feats_queue else 0
    with torch.no_grad():
        for i in range(len_queue):
            img_metas = [each[i] for each in img_metas_list]
            if not img_metas[0]["prev_bev_exists"]:
                prev_bev = None
            feats = [lvl[:, i] for lvl in feats_queue]
            prev_bev = bev_fn(feats, img_metas, prev_bev)
    return prev_bev


class BevHistory:
    """Test-time ``prev_frame_info`` (detectors/bevformer.py:43-50, 236-269): one object per
    video stream.  ``step`` mirrors ``forward_test`` up to the call of ``simple_test``:

      * a new ``scene_token`` drops the stored BEV (:243-247);
      * ``video_test_mode = False`` never uses history (:249-251);
      * the frame's absolute can-bus position ``[:3]`` and yaw ``[-1]`` become deltas against the
        previous frame, or zeros when there is no history (:253-261) — the reference rewrites the
        caller's ``img_metas`` in place; here a deep copy is rewritten and handed on, the caller's
        dict is left alone (``rewritten_metas`` keeps the last one for inspection);
      * the new BEV and the frame's absolute pose are stored for the next call (:265-268)."""

    def __init__(self, video_test_mode=True):
        self.video_test_mode = video_test_mode
        self.prev_frame_info = {"prev_bev": None, "scene_token": None, "prev_pos": 0, "prev_angle": 0}
        self.rewritten_metas = None

    def reset(self):
        self.prev_frame_info = {"prev_bev": None, "scene_token": None, "prev_pos": 0, "prev_angle": 0}

    def step(self, bev_fn, mlvl_feats, img_metas):
        """img_metas: list (batch) of dicts with ``scene_token`` and ``can_bus`` (absolute pose).
        Returns the frame's BEV (bs, Q, C), which is also the next call's history."""
        info = self.prev_frame_info
        if img_metas[0]["scene_token"] != info["scene_token"]:
            info["prev_bev"] = None                    # the first sample of each scene is truncated
        info["scene_token"] = img_metas[0]["scene_token"]
        if not self.video_test_mode:
            info["prev_bev"] = None
        metas = copy.deepcopy(img_metas)
        tmp_pos = copy.deepcopy(metas[0]["can_bus"][:3])
        tmp_angle = copy.deepcopy(metas[0]["can_bus"][-1])
        if info["prev_bev"] is not None:
            metas[0]["can_bus"][:3] -= info["prev_pos"]
            metas[0]["can_bus"][-1] -= info["prev_angle"]
        else:
            metas[0]["can_bus"][-1] = 0
            metas[0]["can_bus"][:3] = 0
        self.rewritten_metas = metas
        with torch.no_grad():
            new_prev_bev = bev_fn(mlvl_feats, metas, info["prev_bev"])
        info["prev_pos"] = tmp_pos
        info["prev_angle"] = tmp_angle
        info["prev_bev"] = new_prev_bev
        return new_prev_bev
