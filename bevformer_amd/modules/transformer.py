"""PerceptionTransformer: the caller of the BEV encoder.

Same registry name (``TRANSFORMER``), constructor arguments, parameters (``level_embeds``,
``cams_embeds``, ``reference_points``, ``can_bus_mlp.*``) and method contracts as
projects/mmdet3d_plugin/bevformer/modules/transformer.py:26-290.  ``get_bev_features``
(:104-200) is the part on the encoder's path and runs on this package's kernels:

  * ego-motion shift: the same float64 numpy arithmetic on ``img_metas[i]['can_bus']``;
  * prev-BEV rotation: ``ops.rotate_bev`` — torchvision's nearest-neighbour ``rotate`` as one
    gather kernel over the (Q, 256) grid (the reference loops over the batch with a
    permute / rotate / permute round trip per sample and overwrites its argument in place;
    this implementation leaves the caller's ``prev_bev`` untouched);
  * camera / level embeddings + flatten: ``ops.flatten_feats`` — one LDS-tiled transpose per
    level instead of a permuted view, two broadcast adds and a ``cat``;
  * can-bus MLP: 18 -> 128 -> 256 on ``bs`` rows, left to torch.

``forward`` (:202-290) additionally needs a decoder; it is built through the
``TRANSFORMER_LAYER_SEQUENCE`` registry when a ``decoder`` config is given (with the
reference plugin imported that is its ``DetectionTransformerDecoder``).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.utils.checkpoint

from .. import ops
from ..registry import (TRANSFORMER, BaseModule, auto_fp16, build_transformer_layer_sequence,
                        xavier_uniform_)
from .spatial_cross_attention import MSDeformableAttention3D
from .temporal_self_attention import TemporalSelfAttention


@TRANSFORMER.register_module(force=True)
class PerceptionTransformer(BaseModule):

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 decoder=None, embed_dims=256, rotate_prev_bev=True, use_shift=True,
                 use_can_bus=True, can_bus_norm=True, use_cams_embeds=True,
                 rotate_center=[100, 100], **kwargs):
        super().__init__(**kwargs)
        self.encoder = build_transformer_layer_sequence(encoder)
        self.decoder = build_transformer_layer_sequence(decoder) if decoder is not None else None
        self.embed_dims = embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.fp16_enabled = False
        self.rotate_prev_bev = rotate_prev_bev
        self.use_shift = use_shift
        self.use_can_bus = use_can_bus
        self.can_bus_norm = can_bus_norm
        self.use_cams_embeds = use_cams_embeds
        self.two_stage_num_proposals = two_stage_num_proposals
        self.init_layers()
        self.rotate_center = rotate_center

    def init_layers(self):
        self.level_embeds = nn.Parameter(torch.Tensor(self.num_feature_levels, self.embed_dims))
        self.cams_embeds = nn.Parameter(torch.Tensor(self.num_cams, self.embed_dims))
        self.reference_points = nn.Linear(self.embed_dims, 3)
        self.can_bus_mlp = nn.Sequential(
            nn.Linear(18, self.embed_dims // 2), nn.ReLU(inplace=True),
            nn.Linear(self.embed_dims // 2, self.embed_dims), nn.ReLU(inplace=True))
        if self.can_bus_norm:
            self.can_bus_mlp.add_module("norm", nn.LayerNorm(self.embed_dims))

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention)) \
                    or type(m).__name__ == "CustomMSDeformableAttention":
                try:
                    m.init_weight()
                except AttributeError:
                    m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)
        xavier_uniform_(self.reference_points, bias=0.0)
        # transformer.py:102 calls mmcv's xavier_init on the can_bus_mlp *Sequential*, which has
        # no .weight / .bias of its own: a no-op (its Linear layers keep the loop's init above)

    @staticmethod
    def bev_shift(img_metas, bev_h, bev_w, grid_length, use_shift):
        """transformer.py:123-141 -> (bs, 2) float64 numpy (x, y)."""
        delta_x = np.array([each["can_bus"][0] for each in img_metas])
        delta_y = np.array([each["can_bus"][1] for each in img_metas])
        ego_angle = np.array([each["can_bus"][-2] / np.pi * 180 for each in img_metas])
        grid_length_y, grid_length_x = grid_length[0], grid_length[1]
        translation_length = np.sqrt(delta_x ** 2 + delta_y ** 2)
        translation_angle = np.arctan2(delta_y, delta_x) / np.pi * 180
        bev_angle = ego_angle - translation_angle
        shift_y = translation_length * np.cos(bev_angle / 180 * np.pi) / grid_length_y / bev_h
        shift_x = translation_length * np.sin(bev_angle / 180 * np.pi) / grid_length_x / bev_w
        return np.stack([shift_x * use_shift, shift_y * use_shift], -1)

    @staticmethod
    def bev_shift_device(can_bus, bev_h, bev_w, grid_length, use_shift):
        """``bev_shift`` for a DEVICE can-bus tensor (bs, 18): the same float64 statements as torch ops —
        no host read, so the whole prologue can sit inside a captured step.  -> (bs, 2) float64."""
        cb = can_bus.to(torch.float64)
        delta_x, delta_y = cb[:, 0], cb[:, 1]
        ego_angle = cb[:, -2] / np.pi * 180
        grid_length_y, grid_length_x = grid_length[0], grid_length[1]
        translation_length = torch.sqrt(delta_x ** 2 + delta_y ** 2)
        translation_angle = torch.atan2(delta_y, delta_x) / np.pi * 180
        bev_angle = ego_angle - translation_angle
        shift_y = translation_length * torch.cos(bev_angle / 180 * np.pi) / grid_length_y / bev_h
        shift_x = translation_length * torch.sin(bev_angle / 180 * np.pi) / grid_length_x / bev_w
        return torch.stack([shift_x * use_shift, shift_y * use_shift], -1)

    @auto_fp16(apply_to=("mlvl_feats", "bev_queries", "prev_bev", "bev_pos"))
    def get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512],
                         bev_pos=None, prev_bev=None, **kwargs):
        """mlvl_feats: list of (bs, Nc, C, h, w); bev_queries (Q, C); bev_pos (bs, C, bev_h,
        bev_w); prev_bev (bs, Q, C) / (Q, bs, C) / None -> bev_embed (bs, Q, C)."""
        low = (torch.float16, torch.bfloat16)
        if any(torch.is_tensor(t) and t.dtype in low for t in (*mlvl_feats, bev_queries, prev_bev, bev_pos)):
            # ``fp16_enabled`` (the decorator above rounded the inputs to half, transformer.py:103): the prologue and
            # encoder kernels are fp32 kernels — widen the rounded inputs once, run with autocast off
            # (BEVFormerEncoder.forward says why that is at least the reference's arithmetic), return fp32
            def up(t):
                return t.float() if torch.is_tensor(t) and t.dtype in low else t
            with torch.autocast(mlvl_feats[0].device.type, enabled=False):
                return self._get_bev_features([up(f) for f in mlvl_feats], up(bev_queries), bev_h, bev_w, grid_length,
                                              up(bev_pos), up(prev_bev), **kwargs)
        return self._get_bev_features(mlvl_feats, bev_queries, bev_h, bev_w, grid_length, bev_pos, prev_bev, **kwargs)

    def _get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length, bev_pos, prev_bev, **kwargs):
        bs = mlvl_feats[0].size(0)
        img_metas = kwargs["img_metas"]
        bev_queries = bev_queries.unsqueeze(1).repeat(1, bs, 1)
        bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        # can_bus given as DEVICE tensors (one (18,) per sample, e.g. views of a static buffer the caller
        # refreshes every frame): shift, rotation and MLP input stay on the device — no host read of the pose
        # anywhere in the step, which makes it capturable in a HIP graph (bevformer_amd.history.GraphedBevHistory)
        dev_pose = all(torch.is_tensor(each["can_bus"]) and each["can_bus"].is_cuda for each in img_metas)
        if dev_pose:
            can_bus_dev = torch.stack([each["can_bus"].reshape(-1) for each in img_metas])
            shift = self.bev_shift_device(can_bus_dev, bev_h, bev_w, grid_length, self.use_shift).to(bev_queries.dtype)
        else:
            shift = bev_queries.new_tensor(self.bev_shift(img_metas, bev_h, bev_w, grid_length,
                                                          self.use_shift))
        if prev_bev is not None:
            if prev_bev.shape[1] == bev_h * bev_w:
                prev_bev = prev_bev.permute(1, 0, 2)
            if self.rotate_prev_bev:
                angles = can_bus_dev[:, -1] if dev_pose else [img_metas[i]["can_bus"][-1] for i in range(bs)]
                prev_bev = ops.rotate_bev(prev_bev, angles, self.rotate_center, bev_h, bev_w)

        if dev_pose:
            can_bus = can_bus_dev.to(bev_queries.dtype)
        else:
            can_bus = bev_queries.new_tensor(np.array([each["can_bus"] for each in img_metas]))
        can_bus = self.can_bus_mlp(can_bus)[None, :, :]
        bev_queries = bev_queries + can_bus * self.use_can_bus

        feat_flatten, spatial_shapes, level_start_index = ops.flatten_feats(
            mlvl_feats, self.cams_embeds if self.use_cams_embeds else None, self.level_embeds)

        return self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w,
                            bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, prev_bev=prev_bev, shift=shift,
                            **kwargs)

    @auto_fp16(apply_to=("mlvl_feats", "bev_queries", "object_query_embed", "prev_bev", "bev_pos"))
    def forward(self, mlvl_feats, bev_queries, object_query_embed, bev_h, bev_w,
                grid_length=[0.512, 0.512], bev_pos=None, reg_branches=None, cls_branches=None,
                prev_bev=None, **kwargs):
        """transformer.py:202-290 -> (bev_embed (Q, bs, C), inter_states, init_reference_out,
        inter_references_out)."""
        if self.decoder is None:
            raise RuntimeError("PerceptionTransformer.forward needs a decoder (built from the "
                               "`decoder` config); get_bev_features does not")
        bev_embed = self.get_bev_features(mlvl_feats, bev_queries, bev_h, bev_w,
                                          grid_length=grid_length, bev_pos=bev_pos,
                                          prev_bev=prev_bev, **kwargs)
        bs = mlvl_feats[0].size(0)
        query_pos, query = torch.split(object_query_embed, self.embed_dims, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        init_reference_out = reference_points
        query = query.permute(1, 0, 2)
        query_pos = query_pos.permute(1, 0, 2)
        bev_embed = bev_embed.permute(1, 0, 2)
        inter_states, inter_references = self.decoder(
            query=query, key=None, value=bev_embed, query_pos=query_pos,
            reference_points=reference_points, reg_branches=reg_branches,
            cls_branches=cls_branches,
            spatial_shapes=torch.tensor([[bev_h, bev_w]], device=query.device),
            level_start_index=torch.tensor([0], device=query.device), **kwargs)
        return bev_embed, inter_states, init_reference_out, inter_references


@TRANSFORMER.register_module(force=True)
class PerceptionTransformerBEVEncoder(BaseModule):
    """The BEVFormerV2 client of the same encoder (modules/transformerV2.py:55-173): camera /
    level embeddings + flatten (``ops.flatten_feats``) and one encoder call without history
    (temporal fusion happens outside, in ``PerceptionTransformerV2``'s ``ResNetFusion``).  The
    training-time BEV-augmentation resampling (:143-172) stays a torch ``grid_sample``."""

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 embed_dims=256, use_cams_embeds=True, rotate_center=[100, 100], **kwargs):
        super().__init__(**kwargs)
        self.encoder = build_transformer_layer_sequence(encoder)
        self.embed_dims = embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.fp16_enabled = False
        self.use_cams_embeds = use_cams_embeds
        self.two_stage_num_proposals = two_stage_num_proposals
        self.rotate_center = rotate_center
        self.level_embeds = nn.Parameter(torch.Tensor(self.num_feature_levels, self.embed_dims))
        if self.use_cams_embeds:
            self.cams_embeds = nn.Parameter(torch.Tensor(self.num_cams, self.embed_dims))

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention)) \
                    or type(m).__name__ == "CustomMSDeformableAttention":
                try:
                    m.init_weight()
                except AttributeError:
                    m.init_weights()
        nn.init.normal_(self.level_embeds)
        if self.use_cams_embeds:
            nn.init.normal_(self.cams_embeds)

    def forward(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512], bev_pos=None,
                prev_bev=None, **kwargs):
        """-> (bs, bev_h * bev_w, C); ``prev_bev`` is accepted and ignored, as in the reference."""
        bs = mlvl_feats[0].size(0)
        bev_queries = bev_queries.unsqueeze(1).repeat(1, bs, 1)
        bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        feat_flatten, spatial_shapes, level_start_index = ops.flatten_feats(
            mlvl_feats, self.cams_embeds if self.use_cams_embeds else None, self.level_embeds)
        bev_embed = self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w,
                                 bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                                 level_start_index=level_start_index, prev_bev=None,
                                 shift=bev_queries.new_tensor([0, 0]).unsqueeze(0), **kwargs)
        prev_bev = bev_embed
        meta0 = kwargs["img_metas"][0]
        if "aug_param" in meta0 and "GlobalRotScaleTransImage_param" in meta0["aug_param"]:
            rot_angle, scale_ratio, flip_dx, flip_dy, bda_mat, only_gt = \
                meta0["aug_param"]["GlobalRotScaleTransImage_param"]
            prev_bev = prev_bev.reshape(bs, bev_h, bev_w, -1).permute(0, 3, 1, 2)
            if only_gt:
                ref_y, ref_x = torch.meshgrid(
                    torch.linspace(0.5, bev_h - 0.5, bev_h, dtype=bev_queries.dtype, device=bev_queries.device),
                    torch.linspace(0.5, bev_w - 0.5, bev_w, dtype=bev_queries.dtype, device=bev_queries.device),
                    indexing="ij")
                grid = torch.stack((ref_x / bev_w, ref_y / bev_h), -1)
                grid_shift = (grid * 2.0 - 1.0).unsqueeze(0).unsqueeze(-1)
                mat = bda_mat[:2, :2].to(grid_shift).view(1, 1, 1, 2, 2).repeat(
                    grid_shift.shape[0], grid_shift.shape[1], grid_shift.shape[2], 1, 1)
                grid_shift = torch.matmul(mat, grid_shift).squeeze(-1)
                prev_bev = torch.nn.functional.grid_sample(prev_bev, grid_shift, align_corners=False)
            prev_bev = prev_bev.reshape(bs, -1, bev_h * bev_w).permute(0, 2, 1)
        return prev_bev


class _BasicBlock(nn.Module):
    """mmdet's ResNet ``BasicBlock`` (third party; used as it is when mmdet is installed): conv3x3 - BN -
    ReLU - conv3x3 - BN, + identity / ``downsample``, ReLU.  Parameter names as in mmdet
    (``conv1, bn1, conv2, bn2, downsample.{0,1}``) so that BEVFormerV2 checkpoints load."""

    def __init__(self, inplanes, planes, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


def _basic_block(inplanes, planes, downsample=None):
    try:                                            # the real one when mmdet is there
        from mmdet.models.backbones.resnet import BasicBlock
        return BasicBlock(inplanes, planes, stride=1, norm_cfg=dict(type="BN"), downsample=downsample)
    except ImportError:
        return _BasicBlock(inplanes, planes, downsample)


class ResNetFusion(BaseModule):
    """Temporal fusion of BEVFormerV2 (modules/transformerV2.py:16-52): the BEV maps of the frames are
    concatenated along channels, passed through ``num_layer`` ResNet basic blocks (3 x 3 convolutions
    over the BEV grid: MIOpen) and projected back to ``out_channels`` by Linear + LayerNorm.  Batch
    norm here is plain ``BatchNorm2d`` (the reference asks for SyncBN: same inference arithmetic)."""

    def __init__(self, in_channels, out_channels, inter_channels, num_layer, norm_cfg=dict(type="SyncBN"),
                 with_cp=False):
        super().__init__()
        layers = []
        self.inter_channels = inter_channels
        for i in range(num_layer):
            if i == 0 and inter_channels != in_channels:
                downsample = nn.Sequential(nn.Conv2d(in_channels, inter_channels, 3, stride=1, padding=1, bias=False),
                                           nn.BatchNorm2d(inter_channels))
                layers.append(_basic_block(in_channels, inter_channels, downsample))
            else:
                layers.append(_basic_block(in_channels if i == 0 else inter_channels, inter_channels))
        self.layers = nn.Sequential(*layers)
        self.layer_norm = nn.Sequential(nn.Linear(inter_channels, out_channels), nn.LayerNorm(out_channels))
        self.with_cp = with_cp

    def forward(self, x):
        """x: list of (bs, C, bev_h, bev_w) -> (bs, bev_h * bev_w, out_channels)."""
        x = torch.cat(x, 1).contiguous()
        for layer in self.layers:
            if self.with_cp and x.requires_grad:
                x = torch.utils.checkpoint.checkpoint(layer, x)
            else:
                x = layer(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        lin, norm = self.layer_norm[0], self.layer_norm[1]
        y = ops.linear_or_torch(x, lin.weight, lin.bias, tag="fusion_proj")
        out = ops.add_layernorm(y, None, norm.weight, norm.bias, norm.eps) \
            if not torch.is_grad_enabled() else None
        return out if out is not None else norm(y)


@TRANSFORMER.register_module(force=True)
class PerceptionTransformerV2(PerceptionTransformerBEVEncoder):
    """BEVFormerV2's transformer (modules/transformerV2.py:177-353): the BEV encoder client above, an
    optional ``ResNetFusion`` over the BEV maps of ``frames`` and the detection decoder."""

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 embed_dims=256, use_cams_embeds=True, rotate_center=[100, 100], frames=(0,), decoder=None,
                 num_fusion=3, inter_channels=None, **kwargs):
        super().__init__(num_feature_levels, num_cams, two_stage_num_proposals, encoder, embed_dims,
                         use_cams_embeds, rotate_center, **kwargs)
        self.decoder = build_transformer_layer_sequence(decoder) if decoder is not None else None
        self.reference_points = nn.Linear(self.embed_dims, 3)
        self.frames = frames
        if len(self.frames) > 1:
            self.fusion = ResNetFusion(len(self.frames) * self.embed_dims, self.embed_dims,
                                       inter_channels if inter_channels is not None
                                       else len(self.frames) * self.embed_dims, num_fusion)

    def init_weights(self):
        super().init_weights()
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention)) \
                    or type(m).__name__ == "CustomMSDeformableAttention":
                try:
                    m.init_weight()
                except AttributeError:
                    m.init_weights()
        nn.init.xavier_uniform_(self.reference_points.weight)
        nn.init.constant_(self.reference_points.bias, 0.0)

    def get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512], bev_pos=None,
                         prev_bev=None, **kwargs):
        return PerceptionTransformerBEVEncoder.forward(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length,
                                                       bev_pos, prev_bev, **kwargs)

    def fuse_frames(self, bev_embed, prev_bev, bev_h, bev_w):
        """transformerV2.py:296-313: slot of frame 0 takes the current BEV, missing earlier frames copy
        their successor and missing later frames their predecessor (detached), then ``fusion``."""
        n, cur = len(self.frames), list(self.frames).index(0)
        assert len(prev_bev) == n and prev_bev[cur] is None
        slots = list(prev_bev)
        slots[cur] = bev_embed

        def known(i, neighbour):
            return slots[i] if slots[i] is not None else slots[neighbour].detach()
        for i in reversed(range(cur)):          # before the current frame: a hole repeats the frame after it
            slots[i] = known(i, i + 1)
        for i in range(cur + 1, n):             # after it: a hole repeats the frame before it
            slots[i] = known(i, i - 1)
        prev_bev[:] = slots                      # (the reference fills the caller's list in place)
        maps = [x.reshape(x.shape[0], bev_h, bev_w, x.shape[-1]).permute(0, 3, 1, 2).contiguous() for x in prev_bev]
        return self.fusion(maps)

    def forward(self, mlvl_feats, bev_queries, object_query_embed, bev_h, bev_w, grid_length=[0.512, 0.512],
                bev_pos=None, reg_branches=None, cls_branches=None, prev_bev=None, **kwargs):
        """-> (bev_embed (Q, bs, C), inter_states, init_reference_out, inter_references_out)."""
        if self.decoder is None:
            raise RuntimeError("PerceptionTransformerV2.forward needs a decoder (built from the `decoder` config)")
        bev_embed = self.get_bev_features(mlvl_feats, bev_queries, bev_h, bev_w, grid_length=grid_length,
                                          bev_pos=bev_pos, prev_bev=None, **kwargs)
        if len(self.frames) > 1:
            bev_embed = self.fuse_frames(bev_embed, prev_bev, bev_h, bev_w)
        bs = mlvl_feats[0].size(0)
        query_pos, query = torch.split(object_query_embed, self.embed_dims, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        init_reference_out = reference_points
        query = query.permute(1, 0, 2)
        query_pos = query_pos.permute(1, 0, 2)
        bev_embed = bev_embed.permute(1, 0, 2)
        inter_states, inter_references = self.decoder(
            query=query, key=None, value=bev_embed, query_pos=query_pos, reference_points=reference_points,
            reg_branches=reg_branches, cls_branches=cls_branches,
            spatial_shapes=torch.tensor([[bev_h, bev_w]], device=query.device),
            level_start_index=torch.tensor([0], device=query.device), **kwargs)
        return bev_embed, inter_states, init_reference_out, inter_references
