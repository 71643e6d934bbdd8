"""PerceptionTransformer.get_bev_features on CPU: the oracle's
restatement against the reference's own transformer.py (build container only), and the host
logic of the product class against the oracle with the kernels routed through the oracle."""
import numpy as np
import pytest
import torch

from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O
from oracle import mmcv_stub

from helpers import build_transformer_pair, oracle_ops, split_transformer_sd


def _oracle_bev(sd, mlvl, bq, kw, name, **flags):
    own, enc = split_transformer_sd(sd)
    w = S.WORKLOADS[name]
    return O.get_bev_features(own, enc, mlvl, bq, kw["bev_h"], kw["bev_w"], bev_pos=kw["bev_pos"],
                              img_metas=kw["img_metas"], pc_range=S.PC_RANGE,
                              grid_length=kw["grid_length"], prev_bev=kw["prev_bev"],
                              rotate_center=(w["bev_w"] // 2, w["bev_h"] // 2), **flags)


@pytest.mark.reference
@pytest.mark.parametrize("name,bs", [("micro", 1), ("micro4", 2)])
@pytest.mark.parametrize("temporal", [False, True])
def test_oracle_prologue_matches_reference_file(name, bs, temporal):
    """The reference's PerceptionTransformer.get_bev_features, unmodified, under the stub
    (torchvision's rotate bound to the oracle's restatement) == the oracle's function."""
    cfg = S.transformer_cfg(name)
    ref = mmcv_stub.build_reference_transformer(
        cfg["encoder"], num_feature_levels=cfg["num_feature_levels"], rotate_center=cfg["rotate_center"])
    torch.manual_seed(1)
    ref.init_weights()
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    own, enc = split_transformer_sd(sd)
    S.trained_like_(enc, seed=3)
    sd.update({"encoder." + k: v for k, v in enc.items()})
    ref.load_state_dict(sd)
    mlvl, bq, kw = S.make_transformer_inputs(name, seed=0, bs=bs, temporal=temporal)
    with torch.no_grad():
        prev = None if kw["prev_bev"] is None else kw["prev_bev"].clone()   # the reference rotates in place
        want = ref.get_bev_features(mlvl, bq, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"],
                                    bev_pos=kw["bev_pos"], prev_bev=prev, img_metas=kw["img_metas"])
        got = _oracle_bev(sd, mlvl, bq, kw, name)
    assert torch.equal(got, want)


@pytest.mark.reference
def test_product_transformer_has_reference_parameters():
    cfg = S.transformer_cfg("micro4")
    ref = mmcv_stub.build_reference_transformer(
        cfg["encoder"], num_feature_levels=cfg["num_feature_levels"], rotate_center=cfg["rotate_center"])
    mine, _ = build_transformer_pair("micro4")
    a, b = mine.state_dict(), ref.state_dict()
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape for k in a)


@pytest.mark.parametrize("name,bs,temporal", [("micro", 1, True), ("micro4", 2, True), ("micro4", 1, False)])
def test_product_prologue_host_logic(name, bs, temporal):
    t, sd = build_transformer_pair(name)
    mlvl, bq, kw = S.make_transformer_inputs(name, seed=1, bs=bs, temporal=temporal)
    prev_before = None if kw["prev_bev"] is None else kw["prev_bev"].clone()
    with torch.no_grad(), oracle_ops():
        got = t.get_bev_features(mlvl, bq, **kw)
        want = _oracle_bev(sd, mlvl, bq, kw, name)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    if temporal:      # unlike the reference, the caller's prev_bev is left untouched
        assert torch.equal(kw["prev_bev"], prev_before)


def test_switches_follow_the_reference_flags():
    t, sd = build_transformer_pair("micro")
    t.use_shift = t.use_can_bus = t.use_cams_embeds = t.rotate_prev_bev = False
    mlvl, bq, kw = S.make_transformer_inputs("micro", seed=2, temporal=True)
    with torch.no_grad(), oracle_ops():
        got = t.get_bev_features(mlvl, bq, **kw)
        want = _oracle_bev(sd, mlvl, bq, kw, "micro", use_shift=False, use_can_bus=False,
                           use_cams_embeds=False, rotate_prev_bev=False)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


def test_rotation_restatement_properties():
    """torchvision is not installed: the restated nearest-neighbour rotate is checked on what
    must hold for any correct implementation — identity at 0 degrees, exact quarter turns about
    the pixel-grid centre, zero fill outside, and inverse consistency of the index map."""
    g = torch.Generator().manual_seed(0)
    img = torch.randn(3, 8, 8, generator=g)
    assert torch.equal(O.rotate_nearest(img, 0.0, [4, 4]), img)
    # centre (4, 4) in pixel-corner coordinates = the exact centre of an 8 x 8 grid:
    # +90 degrees is counter-clockwise in image coordinates (torchvision convention)
    assert torch.equal(O.rotate_nearest(img, 90.0, [4, 4]), torch.rot90(img, 1, (1, 2)))
    assert torch.equal(O.rotate_nearest(img, 180.0, [4, 4]), torch.rot90(img, 2, (1, 2)))
    idx = O.rotate_source_index(12, 10, 33.0, [5, 6])
    assert (idx == -1).any() and (idx >= 0).any() and idx.max() < 120
    out = O.rotate_nearest(torch.ones(1, 12, 10), 33.0, [5, 6])
    assert torch.equal(out.flatten() == 0, idx == -1)


def test_shift_matches_closed_form():
    metas = S.make_img_metas("micro")
    metas[0]["can_bus"] = S.make_can_bus(0)
    sh = O.bev_shift(metas, 12, 10)
    # |t| = 2.5 m, heading 0.3 rad vs motion direction atan2(1.5, 2.0)
    ang = 0.3 - np.arctan2(1.5, 2.0)
    np.testing.assert_allclose(sh[0], [2.5 * np.sin(ang) / 0.512 / 10, 2.5 * np.cos(ang) / 0.512 / 12], rtol=1e-12)


# -- BEVFormerV2's client of the encoder (modules/transformerV2.py:55-173) -----------------------

def _v2_pair(name, seed=11):
    import bevformer_amd
    cfg = S.transformer_cfg(name)
    torch.manual_seed(seed)
    mine = bevformer_amd.build_transformer(dict(type="PerceptionTransformerBEVEncoder",
                                                num_feature_levels=cfg["num_feature_levels"],
                                                encoder=cfg["encoder"])).eval()
    mine.init_weights()
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    own, enc = split_transformer_sd(sd)
    S.trained_like_(enc, seed=3)
    sd.update({"encoder." + k: v for k, v in enc.items()})
    mine.load_state_dict(sd)
    return mine, sd, cfg


@pytest.mark.reference
@pytest.mark.parametrize("aug", [False, True])
def test_v2_bev_encoder_matches_reference_file(aug):
    mine, sd, cfg = _v2_pair("micro4")
    ref = mmcv_stub.build_reference_bev_encoder_v2(cfg["encoder"],
                                                   num_feature_levels=cfg["num_feature_levels"])
    assert list(ref.state_dict()) == list(sd)
    ref.load_state_dict(sd)
    mlvl, bq, kw = S.make_transformer_inputs("micro4", seed=4, bs=1)
    if aug:   # BEV augmentation of BEVFormerV2 training: resample the BEV by a 2 x 2 matrix
        c, s_ = 0.9659258, 0.2588190
        kw["img_metas"][0]["aug_param"] = dict(GlobalRotScaleTransImage_param=(
            15.0, 1.0, False, False, torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]]), True))
    args = dict(grid_length=kw["grid_length"], bev_pos=kw["bev_pos"], img_metas=kw["img_metas"])
    with torch.no_grad():
        want = ref(mlvl, bq, kw["bev_h"], kw["bev_w"], **args)
        with oracle_ops():
            got = mine(mlvl, bq, kw["bev_h"], kw["bev_w"], **args)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


@pytest.mark.reference
@pytest.mark.parametrize("inter", [None, 128])
def test_resnet_fusion_matches_reference_file(inter):
    """ResNetFusion (modules/transformerV2.py:16-52) from the reference's own file — with mmdet's
    BasicBlock restated in the stub (third party, unpinned) — against the product class: same
    parameter names, same output."""
    from bevformer_amd.modules import ResNetFusion
    v2 = mmcv_stub.load_reference_transformer_v2()
    torch.manual_seed(0)
    inc, outc = 2 * 64, 64
    ref = v2.ResNetFusion(inc, outc, inter if inter is not None else inc, 2, norm_cfg=dict(type="BN")).eval()
    mine = ResNetFusion(inc, outc, inter if inter is not None else inc, 2).eval()
    sd = ref.state_dict()
    for k, v in sd.items():                       # non-trivial BN statistics
        if k.endswith("running_var"):
            v.uniform_(0.5, 1.5)
        elif k.endswith("running_mean"):
            v.normal_(0, 0.2)
    assert sorted(mine.state_dict()) == sorted(sd)
    mine.load_state_dict(sd)
    x = [torch.randn(2, 64, 6, 5), torch.randn(2, 64, 6, 5)]
    with torch.no_grad():
        torch.testing.assert_close(mine(x), ref(x), rtol=1e-5, atol=1e-5)


@pytest.mark.reference
def test_transformer_v2_frame_filling_matches_reference_file():
    """PerceptionTransformerV2.forward up to the decoder (transformerV2.py:284-313): current BEV in
    the slot of frame 0, missing neighbours filled from their successor / predecessor, fusion."""
    import bevformer_amd
    v2 = mmcv_stub.load_reference_transformer_v2()
    cfg = S.transformer_cfg("micro")
    kw = dict(num_feature_levels=cfg["num_feature_levels"], encoder=cfg["encoder"], frames=(-1, 0, 1), num_fusion=1,
              inter_channels=64)
    torch.manual_seed(1)
    mine = bevformer_amd.build_transformer(dict(type="PerceptionTransformerV2", **kw)).eval()
    mine.init_weights()
    ref = v2.PerceptionTransformerV2(decoder=dict(type="NullDecoder"), **kw).eval()
    sd = {k: v for k, v in mine.state_dict().items()}
    assert sorted(k for k in ref.state_dict()) == sorted(sd)
    ref.load_state_dict(sd)
    mlvl, bq, tkw = S.make_transformer_inputs("micro", seed=4, bs=1)
    h, w = tkw["bev_h"], tkw["bev_w"]
    args = dict(grid_length=tkw["grid_length"], bev_pos=tkw["bev_pos"], img_metas=tkw["img_metas"])
    later = torch.randn(1, h * w, 256)
    with torch.no_grad():
        with oracle_ops():
            bev = mine.get_bev_features(mlvl, bq, h, w, **args)
            got = mine.fuse_frames(bev, [None, None, later.clone()], h, w)
        bev_r = ref.get_bev_features(mlvl, bq, h, w, **args)
        # the reference's forward body between get_bev_features and the decoder, lines 296-313
        prev = [None, None, later.clone()]
        cur = list(ref.frames).index(0)
        prev[cur] = bev_r
        for i in range(1, cur + 1):
            if prev[cur - i] is None:
                prev[cur - i] = prev[cur - i + 1].detach()
        for i in range(cur + 1, len(ref.frames)):
            if prev[i] is None:
                prev[i] = prev[i - 1].detach()
        want = ref.fusion([x.reshape(x.shape[0], h, w, x.shape[-1]).permute(0, 3, 1, 2).contiguous() for x in prev])
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


def test_rotate_source_index_is_torch_grid_sample_nearest_on_the_torchvision_grid():
    """The inside of the restated ``rotate``: torchvision's tensor path ends in
    ``torch.nn.functional.grid_sample(img, grid, mode='nearest', padding_mode='zeros', align_corners=False)``
    on the grid of ``_gen_affine_grid``.  torch IS installed: run its grid_sample kernel on an index image over the
    same grid and require the oracle's index map (un-normalisation, rounding half to even, bounds) to be the one it
    produces — random angles and centres at the base grid (200 x 200), odd sizes, quarter turns and half-pixel
    centres (which put coordinates exactly on rounding ties)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    cases = [(200, 200, (100, 100), a) for a in (torch.rand(12, generator=g) * 20 - 10).tolist()]
    cases += [(200, 200, (float(cx), float(cy)), a) for cx, cy, a in
              zip((torch.rand(8, generator=g) * 200).tolist(), (torch.rand(8, generator=g) * 200).tolist(),
                  (torch.rand(8, generator=g) * 360 - 180).tolist())]
    cases += [(37, 53, (20, 11), 33.0), (50, 50, (25, 25), 90.0), (50, 50, (24.5, 24.5), 90.0), (12, 10, (5, 6), 180.0),
              (150, 150, (75, 75), -0.37), (200, 200, (100, 100), 0.0), (16, 16, (7.5, 7.5), 45.0)]
    ties = 0
    for h, w, center, angle in cases:
        grid = O.rotate_affine_grid(h, w, angle, list(center))
        img = (torch.arange(h * w, dtype=torch.float32) + 1).view(1, 1, h, w)
        out = F.grid_sample(img, grid.view(1, h, w, 2), mode="nearest", padding_mode="zeros", align_corners=False)
        want = out.view(-1).long() - 1                                    # -1 = zero fill
        got = O.rotate_source_index(h, w, angle, list(center))
        assert torch.equal(got, want), (h, w, center, angle, int((got != want).sum()))
        px = ((grid[:, 0] + 1) * w - 1) / 2
        ties += int(((px - px.floor()) == 0.5).sum())
    assert ties > 0            # the half-pixel-centre quarter turns do sit on ties: the rounding rule was exercised
