"""GPU: CustomMSDeformableAttention / DetectionTransformerDecoder on the
HIP kernels against the CPU oracle.  fp32; tolerances as for the encoder modules (merged GEMM,
fused sampling order): rtol / atol 1e-3 on O(1) outputs after two layers."""
import pytest
import torch

import bevformer_amd
from bevformer_amd import ops
from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O

from test_decoder_cpu import _Reg, _oracle_decoder, _trained

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(params=["split", "native"])
def gemm(request):
    saved = ops.gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(saved)


@pytest.mark.parametrize("bev,nq,bs", [((12, 10), 37, 2), ((50, 50), 900, 1)])
@pytest.mark.parametrize("with_reg", [False, True])
def test_decoder_matches_oracle(gemm, bev, nq, bs, with_reg):
    torch.manual_seed(0)
    dec = bevformer_amd.build_transformer_layer_sequence(S.decoder_cfg(2)).eval()
    sd = _trained(dec.state_dict())
    dec.load_state_dict(sd)
    q, qp, v, ref, shapes, start = S.make_decoder_inputs(*bev, num_query=nq, bs=bs, seed=4)
    reg = _Reg(2) if with_reg else None
    with torch.no_grad():
        want = _oracle_decoder(sd, q, qp, v, ref, shapes, 2, reg)
        dec = dec.to(DEV)
        got = dec(query=q.to(DEV), key=None, value=v.to(DEV), query_pos=qp.to(DEV),
                  reference_points=ref.to(DEV), reg_branches=reg.to(DEV) if reg else None,
                  spatial_shapes=shapes.to(DEV), level_start_index=start.to(DEV))
    torch.testing.assert_close(got[0].cpu(), want[0], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(got[1].cpu(), want[1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("levels,box_refs,batch_first", [(1, False, False), (2, False, True), (1, True, False)])
def test_attention_layouts_and_box_references(levels, box_refs, batch_first):
    torch.manual_seed(0)
    mod = bevformer_amd.registry.build_attention(dict(
        type="CustomMSDeformableAttention", embed_dims=256, num_levels=levels, batch_first=batch_first)).eval()
    sd = _trained({"a." + k: v for k, v in mod.state_dict().items()})
    mod.load_state_dict({k[2:]: v for k, v in sd.items()})
    g = torch.Generator().manual_seed(5)
    shapes = torch.tensor([[12, 10], [6, 5]][:levels], dtype=torch.long)
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    nv, nq, bs = int(shapes.prod(1).sum()), 29, 2
    q, qp, v = (torch.randn(n, bs, 256, generator=g) for n in (nq, nq, nv))
    r = torch.rand(bs, nq, levels, 2, generator=g)
    if box_refs:
        r = torch.cat([r, torch.full_like(r, 0.3)], -1)
    with torch.no_grad():
        want = O.custom_ms_deformable_attention(sd, "a.", q, v, r, shapes, query_pos=qp, num_levels=levels)
        if batch_first:
            q, qp, v = (t.permute(1, 0, 2).contiguous() for t in (q, qp, v))
            want = want.permute(1, 0, 2)
        got = mod.to(DEV)(q.to(DEV), None, v.to(DEV), query_pos=qp.to(DEV), reference_points=r.to(DEV),
                          spatial_shapes=shapes.to(DEV), level_start_index=start.to(DEV)).cpu()
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


def test_attention_backward_matches_oracle_autograd():
    torch.manual_seed(0)
    mod = bevformer_amd.registry.build_attention(dict(type="CustomMSDeformableAttention", embed_dims=256,
                                                      num_levels=1)).eval()
    sd = _trained({"a." + k: v for k, v in mod.state_dict().items()})
    mod.load_state_dict({k[2:]: v for k, v in sd.items()})
    q, qp, v, ref, shapes, start = S.make_decoder_inputs(12, 10, seed=6)
    r = ref[..., :2].unsqueeze(2)
    qd, vd = q.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    out = mod.to(DEV)(qd, None, vd, query_pos=qp.to(DEV), reference_points=r.to(DEV),
                      spatial_shapes=shapes.to(DEV), level_start_index=start.to(DEV))
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(7))
    out.backward(gout.to(DEV))
    sdg = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    qc, vc = q.clone().requires_grad_(True), v.clone().requires_grad_(True)
    O.custom_ms_deformable_attention(sdg, "a.", qc, vc, r, shapes, query_pos=qp).backward(gout)

    def close(a, b, what):
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        assert err < 2e-3, f"{what}: relative max error {err:.3e}"
    close(qd.grad.cpu(), qc.grad, "query grad")
    close(vd.grad.cpu(), vc.grad, "value grad")
    for name, p in mod.named_parameters():
        close(p.grad.cpu(), sdg["a." + name].grad, name)


def test_full_transformer_forward_gpu_matches_cpu_path():
    """PerceptionTransformer.forward built from the reference's transformer config (encoder +
    prologue + 2-layer decoder with self-attention): the GPU product path against the same module
    on CPU with every kernel routed through the oracle."""
    from helpers import oracle_ops
    cfg = S.transformer_cfg("micro")
    cfg["decoder"] = S.reference_decoder_cfg(num_layers=2)
    torch.manual_seed(0)
    t = bevformer_amd.build_transformer(cfg).eval()
    t.init_weights()
    sd = _trained(t.state_dict(), seed=9)
    t.load_state_dict(sd)
    mlvl, bq, kw = S.make_transformer_inputs("micro", seed=0, bs=1, temporal=True)
    oqe = torch.randn(13, 512, generator=torch.Generator().manual_seed(1))
    reg = _Reg(2)
    with torch.no_grad():
        with oracle_ops():
            want = t(mlvl, bq, oqe, reg_branches=reg, **kw)
        kwd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
        got = t.to(DEV)([f.to(DEV) for f in mlvl], bq.to(DEV), oqe.to(DEV), reg_branches=reg.to(DEV), **kwd)
    for g, w_, tol in zip(got, want, (1e-3, 2e-3, 1e-5, 1e-3)):
        torch.testing.assert_close(g.cpu(), w_, rtol=tol, atol=tol)
