"""CPU: the three statements of the operator in oracle/ agree with each other
(grid_sample form = what the reference's CPU fallback computes; scalar loops
from the math; plain C), forward and backward, including out-of-range points."""
import pytest
import torch

from bevformer_amd.synthetic import make_msda_case
from oracle import bevformer_cpu as O
from oracle import msda_c

CASES = [
    # N, Q, M, D, shapes, P
    (1, 5, 2, 4, [(3, 4)], 2),
    (2, 7, 8, 32, [(6, 9), (3, 5)], 4),
    (2, 9, 3, 8, [(5, 7), (3, 4), (2, 2), (1, 1)], 8),
    (1, 4, 1, 5, [(4, 4)], 3),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("coherent", [False, True])
def test_forward_statements_agree(case, coherent):
    N, Q, M, D, shapes, P = case
    value, sh, start, loc, attn = make_msda_case(N, Q, M, D, shapes, P, seed=7, coherent=coherent)
    a = O.msda_gridsample(value, sh, loc, attn)
    b = O.msda_loops(value, sh, start, loc, attn).float()
    c = msda_c.forward(value, sh, start, loc, attn)
    # tolerance: fp32 grid_sample vs float64 loops
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(c, b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", CASES[:3])
def test_backward_c_matches_autograd(case):
    N, Q, M, D, shapes, P = case
    value, sh, start, loc, attn = make_msda_case(N, Q, M, D, shapes, P, seed=11)
    g = torch.randn(N, Q, M * D, generator=torch.Generator().manual_seed(5))
    gv, gl, ga = O.msda_backward_autograd(value.double(), sh, loc.double(), attn.double(), g.double())
    cv, cl, ca = msda_c.backward(value, sh, start, loc, attn, g)
    torch.testing.assert_close(cv, gv.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ca, ga.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(cl, gl.float(), rtol=1e-4, atol=1e-4)


def test_exact_integer_grid_is_identity():
    # sampling exactly at pixel centres with a one-hot weight reproduces value
    H, W, M, D = 3, 5, 2, 4
    value = torch.randn(1, H * W, M, D)
    sh = torch.tensor([[H, W]])
    start = torch.tensor([0])
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    loc = torch.stack([(xs.flatten() + 0.5) / W, (ys.flatten() + 0.5) / H], -1)
    loc = loc.view(1, H * W, 1, 1, 1, 2).expand(1, H * W, M, 1, 1, 2).contiguous()
    attn = torch.ones(1, H * W, M, 1, 1)
    out = msda_c.forward(value, sh, start, loc, attn)
    torch.testing.assert_close(out, value.view(1, H * W, M * D), rtol=0, atol=1e-6)
    torch.testing.assert_close(O.msda_gridsample(value, sh, loc, attn), out, rtol=0, atol=1e-6)


def test_hf_deformable_detr_cross_check():
    """Independent copy of the same published algorithm (HuggingFace
    transformers, Deformable-DETR) agrees with the restated CPU fallback."""
    try:
        from transformers.models.deformable_detr.modeling_deformable_detr import (
            MultiScaleDeformableAttention as HFMSDA)
    except Exception:  # pragma: no cover
        pytest.skip("transformers' deformable_detr not importable")
    value, sh, start, loc, attn = make_msda_case(2, 6, 4, 8, [(5, 6), (3, 3)], 4, seed=3)
    try:
        hf = HFMSDA().forward(value, sh, [tuple(int(v) for v in r) for r in sh], start, loc, attn, 64)
    except Exception:  # signature drift between transformers versions
        pytest.skip("HF MultiScaleDeformableAttention signature differs")
    torch.testing.assert_close(hf, O.msda_gridsample(value, sh, loc, attn), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["micro4", "tiny"])
def test_oracle_rows_helper_equals_the_full_oracle(name):
    """tests/helpers.py::oracle_encoder_rows (the oracle on a subset of BEV queries: what the base-size gradient check
    differentiates) against ``O.encoder_forward`` on the same rows; and EdgeRecorder leaves the operator's output alone."""
    from helpers import EdgeRecorder, build_pair, camera_rows, oracle_encoder_rows
    from bevformer_amd import synthetic as S
    _, sd = build_pair(name)
    q, f, kw = S.make_inputs(name, seed=0, temporal=True)
    rows = torch.randperm(q.shape[0], generator=torch.Generator().manual_seed(3))[: q.shape[0] // 3].sort().values
    with torch.no_grad():
        want = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
        rec = EdgeRecorder(q.shape[0], camera_rows(name, rows), eps=1e-3, query_ids=rows)
        got = oracle_encoder_rows(sd, q, f, rows, pc_range=S.PC_RANGE, msda=rec, **kw)
        full = EdgeRecorder(q.shape[0], camera_rows(name), eps=1e-3)
        again = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, msda=full, **kw)
    torch.testing.assert_close(got, want[:, rows], rtol=1e-5, atol=1e-5)
    assert torch.equal(again, want)
    # the subset run flags exactly the subset's share of the full run's edge-adjacent queries
    assert torch.equal(rec.fragile[rows], full.fragile[rows]) and not rec.fragile[~torch.isin(torch.arange(q.shape[0]), rows)].any()
    assert 0 < int(full.fragile.sum()) < q.shape[0]
