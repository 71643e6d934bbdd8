"""GPU parity of the row-wise helper kernels (csrc/rowops.h) against torch's
own fp32 ops: tolerance rtol 1e-5 / atol 1e-5 (different reduction order)."""
import pytest
import torch
import torch.nn.functional as F

from bevformer_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,C", [(1, 256), (7, 256), (40000, 256), (33, 512), (5, 1024)])
@pytest.mark.parametrize("with_res", [True, False])
def test_add_layernorm(rows, C, with_res):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 0.5).to(DEV)
    res = torch.randn(rows, C, generator=g).to(DEV) if with_res else None
    w = torch.randn(C, generator=g).to(DEV)
    b = torch.randn(C, generator=g).to(DEV)
    got = ops.add_layernorm(x, res, w, b, 1e-5)
    want = F.layer_norm(x + res if with_res else x, (C,), w, b, 1e-5)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # 3-D input keeps its shape
    if rows % 1 == 0 and rows > 1:
        got3 = ops.add_layernorm(x.view(1, rows, C), None if res is None else res.view(1, rows, C), w, b, 1e-5)
        assert got3.shape == (1, rows, C)
        torch.testing.assert_close(got3.view(rows, C), want, rtol=1e-5, atol=1e-5)


def test_add_layernorm_declines_other_widths():
    x = torch.randn(4, 192, device=DEV)
    assert ops.add_layernorm(x, None, torch.ones(192, device=DEV), torch.zeros(192, device=DEV), 1e-5) is None


@pytest.mark.parametrize("Q,J,R,C", [(1, 1, 1, 256), (50, 3, 90, 256), (4000, 4, 6000, 64)])
def test_gather_mean(Q, J, R, C):
    g = torch.Generator().manual_seed(Q + J)
    rows = torch.randn(R, C, generator=g)
    idx = torch.randint(-1, R, (Q, J), generator=g, dtype=torch.int32)
    scale = torch.rand(Q, generator=g)
    want = torch.zeros(Q, C)
    for j in range(J):
        ok = idx[:, j] >= 0
        want[ok] += rows[idx[ok, j].long()]
    want *= scale[:, None]
    got = ops.gather_mean(rows.to(DEV), idx.to(DEV), scale.to(DEV)).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_ffn_inference_path_matches_sequential():
    from bevformer_amd.modules.bricks import FFN
    torch.manual_seed(0)
    ffn = FFN(embed_dims=256, feedforward_channels=512).to(DEV).eval()
    x = torch.randn(2, 333, 256, device=DEV)
    with torch.no_grad():
        fast = ffn(x)
        slow = x + ffn.layers(x)
    torch.testing.assert_close(fast, slow, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,C", [(1, 256), (37, 256), (40000, 256), (5003, 512)])
def test_add_layernorm_autograd_matches_torch(rows, C):
    """``ops.add_layernorm_autograd`` (forward row kernel + ``bevmsda_add_layernorm_backward_f32``) against
    torch's add + LayerNorm under autograd: output, the (shared) gradient of both addends, weight / bias
    gradients (column sums over all rows through per-block partials + atomics)."""
    from bevformer_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(DEV)
    res = torch.randn(rows, C, generator=g).to(DEV)
    norm = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g) * 0.3 + 1.0)
        norm.bias.copy_(torch.randn(C, generator=g) * 0.2)
    gout = torch.randn(rows, C, generator=g).to(DEV)
    x1, r1 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = ops.add_layernorm_autograd(x1, r1, norm)
    assert y is not None
    y.backward(gout)
    got = (y.detach(), x1.grad.clone(), r1.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone())
    norm.weight.grad = norm.bias.grad = None
    x2, r2 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    want = norm(x2 + r2)
    want.backward(gout)
    torch.testing.assert_close(got[0], want.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[1], x2.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got[2], r2.grad, rtol=1e-4, atol=1e-5)
    tol = dict(rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5))
    torch.testing.assert_close(got[3], norm.weight.grad, **tol)
    torch.testing.assert_close(got[4], norm.bias.grad, **tol)
