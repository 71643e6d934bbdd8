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
import os

import torch


def obtain_history_bev(bev_fn, feats_queue, img_metas_list, modules=()):
    """detectors/bevformer.py:158-177.

    ``modules``: the ``nn.Module`` objects behind ``bev_fn`` (e.g. the ``PerceptionTransformer``).  The reference
    brackets the history frames with ``self.eval()`` / ``self.train()`` (:163, :176): dropout (p = 0.1 in the
    TSA / SCA / FFN of the BEVFormer configs) must be inactive while the history BEVs are computed.  The modules
    given here are switched to eval mode for the walk and put back into the mode they were in (the reference
    calls ``train()`` unconditionally: it only runs this from ``forward_train``); with ``modules=()`` the caller
    owns the mode — history frames computed by a module in training mode differ from the reference's.

    ``feats_queue``: per level a tensor (bs, len_queue, Nc, C, h, w) — the backbone features of
    the queue's earlier frames (the reference slices ``each_scale[:, i]``, :173);
    ``img_metas_list``: per batch element a list / dict of per-frame metas (``each[i]``, :170).
    Frames are walked in order without gradients; a frame whose meta says
    ``prev_bev_exists == False`` starts a new scene (:171-172).  Returns the last BEV or None
    for an empty queue."""
    prev_bev = None
    len_queue = feats_queue[0].shape[1] if feats_queue else 0
    was_training = [m.training for m in modules]
    for m in modules:
        m.eval()
    try:
        with torch.no_grad():
            for i in range(len_queue):
                img_metas = [each[i] for each in img_metas_list]
                if not img_metas[0]["prev_bev_exists"]:
                    prev_bev = None
                feats = [lvl[:, i] for lvl in feats_queue]
                prev_bev = bev_fn(feats, img_metas, prev_bev)
    finally:
        for m, t in zip(modules, was_training):
            m.train(t)
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


class GraphedBevHistory(BevHistory):
    """``BevHistory`` whose per-frame device work — the prologue of ``get_bev_features`` (pose -> shift,
    rotation of the history BEV, can-bus MLP; camera-feature flatten), the frame plan and the encoder — is
    replayed from two captured HIP graphs (a scene's first frame / a frame with history).

    The host keeps running the reference's state machine (scene reset, absolute pose -> delta,
    detectors/bevformer.py:243-268) and refreshes two small DEVICE tensors per frame — the rewritten can-bus
    vector (float64, as the reference's numpy arithmetic) and the camera matrices — which the captured kernels
    read: ``PerceptionTransformer.get_bev_features`` takes the pose from device tensors
    (``bev_shift_device``, ``ops.rotation_theta_device``, ``bevmsda_rotate_bev_dev_f32``) and the encoder plans
    the frame on the device (``csrc/frame_plan.h``), so nothing in the step reads the pose on the host.

    ``bev_fn(mlvl_feats, img_metas, prev_bev)`` as for ``BevHistory``; ``mlvl_feats`` are STATIC buffers (the
    backbone writes each frame's features into them; ``step`` copies when handed other tensors).  The
    returned BEV is a static buffer too: it is overwritten by the next ``step``.

    ``overlap_value_proj``: the encoder's two-stream form of the hoisted camera-value projection (modes.overlap_value_proj)
    inside the captured frames.  Off by default HERE although it is on for a bare encoder step: a frame's kernels do finish
    0.12 ms earlier with it (3.93 against 4.05 ms at base), but a two-queue graph that alternates with stream work (the
    pose copies) pays cross-queue signals at every graph boundary — 17.31 against 17.08 ms per four base frames, 12.95
    against 12.59 with bf16 arithmetic (``tools/history_staging_ab.sh``; None / BEVMSDA_QUEUE_OVERLAP=1 turn it on)."""

    STAGING_SLOTS = int(os.environ.get("BEVMSDA_STAGING_SLOTS", "8"))

    def __init__(self, bev_fn, mlvl_feats, video_test_mode=True, overlap_value_proj=None):
        super().__init__(video_test_mode)
        self.bev_fn = bev_fn
        if overlap_value_proj is None:
            overlap_value_proj = os.environ.get("BEVMSDA_QUEUE_OVERLAP", "0") == "1"
        self.overlap_value_proj = bool(overlap_value_proj)
        self.feats = list(mlvl_feats)
        self.device = self.feats[0].device
        self.graphs = {}
        self.can_bus = None            # (bs, 18) float64, device
        self.l2i = None                # (bs, Nc, 4, 4) float32, device
        self.static_metas = None
        self.prev = None               # (bs, Q, C): the history the next frame reads
        self.out = {}
        self._staging = []             # pinned host slots of the per-frame can-bus / camera-matrix copies
        self._staged = 0

    def _static_inputs(self, metas):
        import numpy as np
        bs = len(metas)
        cb = torch.tensor(np.array([np.asarray(m["can_bus"], dtype=np.float64) for m in metas]), dtype=torch.float64)
        l2i = torch.tensor(np.array([np.asarray(m["lidar2img"], dtype=np.float64) for m in metas]), dtype=torch.float32)
        # what the captured graphs were built for: later frames must bring the same batch size, camera count and
        # image shape (static buffers and kernel arguments were sized / baked from the first frame)
        sig = (bs, tuple(l2i.shape), tuple(tuple(s) for s in metas[0].get("img_shape", ())))
        if getattr(self, "_capture_sig", None) is None:
            self._capture_sig = sig
        elif sig != self._capture_sig:
            raise RuntimeError(f"GraphedBevHistory: frame signature {sig} differs from the captured one "
                               f"{self._capture_sig}; build a new GraphedBevHistory for another rig / batch size")
        if self.can_bus is None:
            self.can_bus = cb.to(self.device)
            self.l2i = l2i.to(self.device)
            self.static_metas = []
            for i in range(bs):
                m = {k: v for k, v in metas[i].items() if k not in ("can_bus", "lidar2img")}
                m["can_bus"] = self.can_bus[i]
                m["lidar2img"] = self.l2i[i]
                self.static_metas.append(m)
        else:
            # two small host -> device copies per frame (144 + 384 bytes per sample), asynchronous: out of pageable memory a
            # copy blocks the host until the stream reaches it, i.e. until the previous frame's graph has finished, and every
            # frame then pays its own launch latency.  A ring of pinned staging slots lets the host run ahead of the GPU; a
            # slot is reused only after the copy that read it has executed.
            if self.STAGING_SLOTS <= 0:           # (A/B: the blocking copies)
                self.can_bus.copy_(cb)
                self.l2i.copy_(l2i)
                return
            if not self._staging:
                self._staging = [dict(cb=torch.empty_like(cb).pin_memory(), l2i=torch.empty_like(l2i).pin_memory(), done=None)
                                 for _ in range(self.STAGING_SLOTS)]
            slot = self._staging[self._staged % len(self._staging)]
            self._staged += 1
            if slot["done"] is not None:
                slot["done"].synchronize()
            slot["cb"].copy_(cb)
            slot["l2i"].copy_(l2i)
            self.can_bus.copy_(slot["cb"], non_blocking=True)
            self.l2i.copy_(slot["l2i"], non_blocking=True)
            slot["done"] = torch.cuda.Event()
            slot["done"].record(torch.cuda.current_stream(self.device))

    def _run(self, has_prev):
        from . import modes
        with modes.using(overlap_value_proj=self.overlap_value_proj):
            out = self.bev_fn(self.feats, self.static_metas, self.prev if has_prev else None)
        if self.prev is None:
            self.prev = torch.empty_like(out)
        self.prev.copy_(out)           # the next frame's history (inside the captured step)
        return out

    def step(self, bev_fn, mlvl_feats, img_metas):
        """Same contract as ``BevHistory.step`` (``bev_fn`` is ignored: the one given at construction is what
        the graphs hold)."""
        info = self.prev_frame_info
        if img_metas[0]["scene_token"] != info["scene_token"]:
            info["prev_bev"] = None
        info["scene_token"] = img_metas[0]["scene_token"]
        if not self.video_test_mode:
            info["prev_bev"] = None
        metas = copy.deepcopy(img_metas)
        tmp_pos = copy.deepcopy(metas[0]["can_bus"][:3])
        tmp_angle = copy.deepcopy(metas[0]["can_bus"][-1])
        has_prev = info["prev_bev"] is not None
        if has_prev:
            metas[0]["can_bus"][:3] -= info["prev_pos"]
            metas[0]["can_bus"][-1] -= info["prev_angle"]
        else:
            metas[0]["can_bus"][-1] = 0
            metas[0]["can_bus"][:3] = 0
        self.rewritten_metas = metas
        for dst, src in zip(self.feats, mlvl_feats):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self._static_inputs(metas)
        with torch.no_grad():
            graph = self.graphs.get(has_prev)
            if graph is None:
                # first frame of this kind: two eager runs (planners, weight images, allocator state) and the
                # capture; the eager runs advance the stored history, so it is put back before the replay
                saved = self.prev.clone() if has_prev else None
                self._run(has_prev)
                side = torch.cuda.Stream(self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    if has_prev:
                        self.prev.copy_(saved)
                    self._run(has_prev)
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.out[has_prev] = self._run(has_prev)
                self.graphs[has_prev] = graph
                if has_prev:
                    self.prev.copy_(saved)
            graph.replay()
        info["prev_pos"] = tmp_pos
        info["prev_angle"] = tmp_angle
        info["prev_bev"] = self.prev
        return self.out[has_prev]
