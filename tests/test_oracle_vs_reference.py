"""Build container only: the oracle restatement and the product modules against
the reference's OWN files executed under the mmcv stub (oracle/mmcv_stub.py).
Skipped automatically where /root/reference does not exist (the GPU box)."""
import pytest
import torch

from bevformer_amd import synthetic as S
from oracle import bevformer_cpu as O
from oracle import mmcv_stub

from helpers import build_pair, oracle_ops

pytestmark = pytest.mark.reference


@pytest.mark.parametrize("name", ["micro", "micro4", "tiny"])
@pytest.mark.parametrize("temporal", [False, True])
def test_restatement_is_bit_exact(name, temporal):
    ref = mmcv_stub.build_reference_encoder(S.encoder_cfg(name))
    sd = S.trained_like_({k: v.clone() for k, v in ref.state_dict().items()}, seed=3)
    ref.load_state_dict(sd)
    q, f, kw = S.make_inputs(name, seed=0, temporal=temporal)
    with torch.no_grad():
        want = ref(q, f, f, **kw)
        got = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
    assert torch.equal(got, want)


def test_reference_init_is_reproduced_bitwise():
    import bevformer_amd
    torch.manual_seed(0)
    mine = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg("tiny"))
    torch.manual_seed(0)
    ref = mmcv_stub.build_reference_encoder(S.encoder_cfg("tiny"))
    a, b = mine.state_dict(), ref.state_dict()
    assert list(a) == list(b)
    assert all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("temporal", [False, True])
def test_product_modules_load_reference_state_dict(temporal):
    """Reference-initialised (degenerate) AND trained-like weights, loaded from
    the reference module's state_dict, give the reference's output."""
    for regime in ("init", "trained"):
        ref = mmcv_stub.build_reference_encoder(S.encoder_cfg("micro4"))
        if regime == "trained":
            ref.load_state_dict(S.trained_like_(
                {k: v.clone() for k, v in ref.state_dict().items()}, seed=5))
        enc, _ = build_pair("micro4")
        enc.load_state_dict(ref.state_dict())
        q, f, kw = S.make_inputs("micro4", seed=2, temporal=temporal)
        with torch.no_grad():
            want = ref(q, f, f, **kw)
            with oracle_ops():
                got = enc(q, f, f, **kw)
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


def test_reference_function_calls_our_ext_shape():
    """The reference's own autograd Function binds whatever ``load_ext``
    returns; check it calls an ``_ext``-shaped object with exactly the argument
    lists bevformer_amd/ext.py implements (signature-level drop-in proof; the
    numerical one runs on the GPU box with the product modules)."""
    import inspect
    from bevformer_amd import ext
    calls = []

    class Spy:
        def ms_deform_attn_forward(self, *a, **k):
            calls.append(("fwd", len(a), sorted(k)))
            inspect.signature(ext.ms_deform_attn_forward).bind(*a, **k)
            return O.msda_gridsample(a[0], a[1], a[3], a[4])

        def ms_deform_attn_backward(self, *a, **k):
            calls.append(("bwd", len(a), sorted(k)))
            inspect.signature(ext.ms_deform_attn_backward).bind(*a, **k)

    ns = mmcv_stub.load_reference(ext_module=Spy())
    value, sh, start, loc, attn = S.make_msda_case(1, 5, 2, 4, [(3, 4)], 2, seed=0)
    value.requires_grad_(True)
    out = ns.MultiScaleDeformableAttnFunction_fp32.apply(value, sh, start, loc, attn, 64)
    out.sum().backward()
    assert calls == [("fwd", 5, ["im2col_step"]), ("bwd", 9, ["im2col_step"])]
