"""TEST INFRASTRUCTURE — regenerates tests/golden/*.pt (build container only).

Every fixture is produced by the REFERENCE'S OWN FILES
(/root/reference/projects/mmdet3d_plugin/bevformer/modules/{encoder,
spatial_cross_attention,temporal_self_attention,custom_base_transformer_layer,
multi_scale_deformable_attn_function,transformer}.py) executed unmodified on CPU under
oracle/mmcv_stub.py, on the seeded synthetic inputs of bevformer_amd/synthetic.py.
Inputs and weights are NOT stored (they are regenerated from the seeds; their
checksums are stored and verified by the tests), outputs are.

    python -m oracle.make_golden
"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bevformer_amd import synthetic as S  # noqa: E402
from oracle import mmcv_stub  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ENCODER_CASES = [("micro", False), ("micro", True), ("micro4", False), ("micro4", True),
                 ("tiny", True)]
TRANSFORMER_CASES = [("micro4", 1), ("micro", 2)]
WEIGHT_SEED, INPUT_SEED = 3, 0


def checksum(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def reference_state_dict(name):
    """Reference-initialised encoder (seed 0) with trained-like weights."""
    torch.manual_seed(0)
    ref = mmcv_stub.build_reference_encoder(S.encoder_cfg(name))
    sd = S.trained_like_({k: v.clone() for k, v in ref.state_dict().items()}, seed=WEIGHT_SEED)
    ref.load_state_dict(sd)
    return ref, sd


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, temporal in ENCODER_CASES:
        ref, sd = reference_state_dict(name)
        q, f, kw = S.make_inputs(name, seed=INPUT_SEED, temporal=temporal)
        with torch.no_grad():
            out = ref(q, f, f, **kw)
        ins = [q, f, kw["bev_pos"], kw["shift"]] + ([kw["prev_bev"]] if temporal else [])
        blob = dict(workload=name, temporal=temporal, weight_seed=WEIGHT_SEED,
                    input_seed=INPUT_SEED, output=out.clone(),
                    input_sha256=checksum(ins), weights_sha256=checksum([sd[k] for k in sorted(sd)]),
                    producer="reference files under oracle/mmcv_stub.py", torch=torch.__version__)
        path = os.path.join(OUT, f"encoder_{name}_{'hist' if temporal else 'first'}.pt")
        torch.save(blob, path)
        print(path, tuple(out.shape), os.path.getsize(path))

    # the encoder's caller: the reference's PerceptionTransformer.get_bev_features
    # (modules/transformer.py:104-200, unmodified; torchvision's rotate = the oracle's
    # restatement, see the header of oracle/bevformer_cpu.py).  The encoder inside gets the
    # same seeded weights as the encoder fixtures above; the transformer's own parameters
    # (embeddings, can-bus MLP: 0.2 MB) are stored.
    for name, bs in TRANSFORMER_CASES:
        cfg = S.transformer_cfg(name)
        ref = mmcv_stub.build_reference_transformer(
            cfg["encoder"], num_feature_levels=cfg["num_feature_levels"],
            rotate_center=cfg["rotate_center"])
        torch.manual_seed(1)
        ref.init_weights()
        _, enc_sd = reference_state_dict(name)
        ref.encoder.load_state_dict(enc_sd)
        own = {k: v.clone() for k, v in ref.state_dict().items()
               if not k.startswith(("encoder.", "decoder."))}
        mlvl, bq, kw = S.make_transformer_inputs(name, seed=INPUT_SEED, bs=bs, temporal=True)
        with torch.no_grad():
            out = ref.get_bev_features(mlvl, bq, kw["bev_h"], kw["bev_w"], grid_length=kw["grid_length"],
                                       bev_pos=kw["bev_pos"], prev_bev=kw["prev_bev"].clone(),
                                       img_metas=kw["img_metas"])
        blob = dict(workload=name, bs=bs, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED,
                    output=out.clone(), own_parameters=own,
                    input_sha256=checksum(mlvl + [bq, kw["bev_pos"], kw["prev_bev"]]),
                    weights_sha256=checksum([enc_sd[k] for k in sorted(enc_sd)]),
                    producer="reference transformer.py + encoder files under oracle/mmcv_stub.py",
                    torch=torch.__version__)
        path = os.path.join(OUT, f"bev_features_{name}_bs{bs}.pt")
        torch.save(blob, path)
        print(path, tuple(out.shape), os.path.getsize(path))

    # operator level: the reference's autograd Function cannot run without its CUDA
    # ext, so these come from the reference MODULES' CPU branch
    # (spatial_cross_attention.py:393-395) = the stub's restated fallback + autograd.
    from oracle import bevformer_cpu as O
    cases = {}
    for i, (N, Q, M, D, shapes, P) in enumerate([(2, 33, 8, 32, [(6, 9), (3, 5)], 4),
                                                 (2, 40, 8, 32, [(8, 13), (4, 7), (2, 4), (1, 2)], 8),
                                                 (2, 60, 8, 32, [(12, 10)], 4)]):
        value, sh, start, loc, attn = S.make_msda_case(N, Q, M, D, shapes, P, seed=100 + i)
        g = torch.randn(N, Q, M * D, generator=torch.Generator().manual_seed(200 + i))
        out = O.msda_gridsample(value, sh, loc, attn)
        gv, gl, ga = O.msda_backward_autograd(value, sh, loc, attn, g)
        cases[f"case{i}"] = dict(dims=(N, Q, M, D, shapes, P), seed=100 + i, gseed=200 + i,
                                 out=out, grad_value=gv, grad_loc=gl, grad_attn=ga)
    path = os.path.join(OUT, "msda_operator.pt")
    torch.save(cases, path)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
