"""Decoder-side classes on CPU: the oracle's restatement against the
reference's own decoder.py executed under the stub (build container only), and the host logic
of the product classes against the oracle with the operator routed through the oracle."""
import copy

import pytest
import torch

import bevformer_amd
from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O
from oracle import mmcv_stub

from helpers import oracle_ops

H, W = 12, 10


def _trained(sd, seed=7):
    return S.trained_like_({k: v.clone() for k, v in sd.items()}, seed=seed)


def _oracle_decoder(sd, q, qp, v, ref, shapes, num_layers, reg=None):
    def layer(i):
        pre = f"layers.{i}."

        def fn(x, ref_in):
            x = O.custom_ms_deformable_attention(sd, pre + "attentions.0.", x, v, ref_in, shapes,
                                                 query_pos=qp)
            x = O.layer_norm(sd, pre + "norms.0", x)
            x = O.ffn(sd, pre + "ffns.0.", x)
            return O.layer_norm(sd, pre + "norms.1", x)
        return fn
    return O.detection_decoder([layer(i) for i in range(num_layers)], q, ref, reg, True)


class _Reg(torch.nn.Module):
    """Stand-in for the head's reg_branches (dense_heads/bevformer_head.py:80-97): 256 -> 10."""

    def __init__(self, n, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.branches = torch.nn.ModuleList([torch.nn.Linear(256, 10) for _ in range(n)])

    def __getitem__(self, i):
        return self.branches[i]


@pytest.mark.reference
@pytest.mark.parametrize("box_refs", [False, True])
def test_attention_restatement_is_bit_exact(box_refs):
    ns = mmcv_stub.load_reference_decoder()
    ref_mod = ns.CustomMSDeformableAttention(embed_dims=256, num_levels=1).eval()
    sd = _trained({"a." + k: v for k, v in ref_mod.state_dict().items()})
    ref_mod.load_state_dict({k[2:]: v for k, v in sd.items()})
    q, qp, v, ref, shapes, start = S.make_decoder_inputs(H, W, seed=1)
    r = ref[..., :2].unsqueeze(2)
    if box_refs:
        r = torch.cat([r, torch.full_like(r, 0.2)], -1)
    with torch.no_grad():
        want = ref_mod(q, None, v, query_pos=qp, reference_points=r, spatial_shapes=shapes,
                       level_start_index=start)
        got = O.custom_ms_deformable_attention(sd, "a.", q, v, r, shapes, query_pos=qp)
    assert torch.equal(got, want)


@pytest.mark.reference
@pytest.mark.parametrize("with_reg", [False, True])
def test_decoder_loop_restatement_is_bit_exact(with_reg):
    ns = mmcv_stub.load_reference_decoder()
    torch.manual_seed(0)
    dec = ns.build_transformer_layer_sequence(copy.deepcopy(S.decoder_cfg(2))).eval()
    sd = _trained(dec.state_dict())
    dec.load_state_dict(sd)
    q, qp, v, ref, shapes, start = S.make_decoder_inputs(H, W, seed=2)
    reg = _Reg(2) if with_reg else None
    with torch.no_grad():
        want = dec(query=q, key=None, value=v, query_pos=qp, reference_points=ref, reg_branches=reg,
                   spatial_shapes=shapes, level_start_index=start)
        got = _oracle_decoder(sd, q, qp, v, ref, shapes, 2, reg)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(ns.inverse_sigmoid(ref), O.inverse_sigmoid(ref))


@pytest.mark.reference
def test_product_decoder_has_reference_parameters_and_init():
    ns = mmcv_stub.load_reference_decoder()
    torch.manual_seed(0)
    want = ns.build_transformer_layer_sequence(copy.deepcopy(S.decoder_cfg(2))).state_dict()
    torch.manual_seed(0)
    got = bevformer_amd.build_transformer_layer_sequence(S.decoder_cfg(2)).state_dict()
    assert list(got) == list(want)
    assert all(torch.equal(got[k], want[k]) for k in got)


@pytest.mark.parametrize("with_reg", [False, True])
def test_product_decoder_host_logic(with_reg):
    torch.manual_seed(0)
    dec = bevformer_amd.build_transformer_layer_sequence(S.decoder_cfg(2)).eval()
    sd = _trained(dec.state_dict())
    dec.load_state_dict(sd)
    q, qp, v, ref, shapes, start = S.make_decoder_inputs(H, W, seed=3)
    reg = _Reg(2) if with_reg else None
    with torch.no_grad(), oracle_ops():
        got = dec(query=q, key=None, value=v, query_pos=qp, reference_points=ref, reg_branches=reg,
                  spatial_shapes=shapes, level_start_index=start)
        want = _oracle_decoder(sd, q, qp, v, ref, shapes, 2, reg)
    torch.testing.assert_close(got[0], want[0], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(got[1], want[1], rtol=1e-5, atol=1e-5)
    assert got[0].shape == (2, 37, 2, 256) and got[1].shape == (2, 2, 37, 3)


def test_inverse_sigmoid_clamps():
    from bevformer_amd.modules.decoder import inverse_sigmoid
    x = torch.tensor([-1.0, 0.0, 0.25, 1.0, 2.0])
    y = inverse_sigmoid(x)
    assert torch.isfinite(y).all() and torch.allclose(y[2], torch.log(torch.tensor(1 / 3.0)))
    assert torch.equal(y, O.inverse_sigmoid(x))


# -- the reference's full decoder / transformer config, stand-alone ---------------------------------

def test_multihead_attention_wrapper_semantics():
    """mmcv's wrapper restated (third-party, unpinned): residual, positional encodings added to
    query AND key, value left alone, parameters under ``attn.*``."""
    from bevformer_amd.modules.decoder import MultiheadAttention
    torch.manual_seed(0)
    m = MultiheadAttention(embed_dims=32, num_heads=4, dropout=0.1).eval()
    assert sorted(m.state_dict()) == ["attn.in_proj_bias", "attn.in_proj_weight", "attn.out_proj.bias",
                                      "attn.out_proj.weight"]
    q, pos = torch.randn(7, 2, 32), torch.randn(7, 2, 32)
    with torch.no_grad():
        got = m(q, query_pos=pos)
        want = q + m.attn(q + pos, q + pos, q)[0]
    torch.testing.assert_close(got, want)
    mb = MultiheadAttention(embed_dims=32, num_heads=4, batch_first=True).eval()
    mb.load_state_dict(m.state_dict())
    with torch.no_grad():
        torch.testing.assert_close(mb(q.transpose(0, 1), query_pos=pos.transpose(0, 1)).transpose(0, 1), got)


def test_reference_decoder_config_builds_and_transformer_forward_runs():
    """bevformer_base.py's decoder block builds from this package's registries and
    PerceptionTransformer.forward (transformer.py:202-290) runs end to end (CPU, operator routed
    through the oracle): shapes and reference-point refinement as the head expects them."""
    cfg = S.transformer_cfg("micro")
    cfg["decoder"] = S.reference_decoder_cfg(num_layers=2)
    torch.manual_seed(0)
    t = bevformer_amd.build_transformer(cfg).eval()
    t.init_weights()
    keys = list(t.state_dict())
    assert "decoder.layers.0.attentions.0.attn.in_proj_weight" in keys
    assert "decoder.layers.1.attentions.1.sampling_offsets.weight" in keys
    mlvl, bq, kw = S.make_transformer_inputs("micro", seed=0, bs=2, temporal=True)
    nq = 11
    object_query_embed = torch.randn(nq, 512, generator=torch.Generator().manual_seed(1))
    reg = _Reg(2)
    with torch.no_grad(), oracle_ops():
        bev, inter, init_ref, inter_ref = t(mlvl, bq, object_query_embed, reg_branches=reg, **kw)
    Q = kw["bev_h"] * kw["bev_w"]
    assert bev.shape == (Q, 2, 256) and inter.shape == (2, nq, 2, 256)
    assert init_ref.shape == (2, nq, 3) and inter_ref.shape == (2, 2, nq, 3)
    assert torch.isfinite(inter).all() and ((inter_ref > 0) & (inter_ref < 1)).all()
    assert not torch.equal(inter_ref[0], init_ref)        # refinement moved the reference points
