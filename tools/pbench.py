"""Micro-benchmark of the get_bev_features prologue kernels at bevformer_base sizes (GPU box):
prev-BEV rotation and camera-feature flatten + embeddings, HIP kernels vs the torch statements
of the reference lines (transformer.py:146-156, :165-184), and the whole
PerceptionTransformer.get_bev_features call next to the bare encoder call.

    python tools/pbench.py [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_amd  # noqa: E402
from bevformer_amd import ops  # noqa: E402
from bevformer_amd import synthetic as S  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


def torch_flatten(mlvl, ce, le):
    flat = []
    for lvl, feat in enumerate(mlvl):
        feat = feat.flatten(3).permute(1, 0, 3, 2)
        feat = feat + ce[:, None, None, :]
        feat = feat + le[None, None, lvl:lvl + 1, :]
        flat.append(feat)
    return torch.cat(flat, 2).permute(0, 2, 1, 3)


def rotate_index(h, w, angle, center):
    """Index map of the torch statement of the rotation (same arithmetic as ops.rotation_theta)."""
    theta = torch.tensor(ops.rotation_theta(angle, center, h, w)).reshape(2, 3)
    xs = torch.arange(w, dtype=torch.float32) + (0.5 - 0.5 * w)
    ys = torch.arange(h, dtype=torch.float32) + (0.5 - 0.5 * h)
    gx = xs[None, :] * theta[0, 0] + ys[:, None] * theta[0, 1] + theta[0, 2]
    gy = xs[None, :] * theta[1, 0] + ys[:, None] * theta[1, 1] + theta[1, 2]
    ix = torch.round(((gx + 1) * w - 1) / 2)
    iy = torch.round(((gy + 1) * h - 1) / 2)
    ok = (ix >= 0) & (ix <= w - 1) & (iy >= 0) & (iy <= h - 1)
    return torch.where(ok, (iy * w + ix).long(), torch.full((h, w), -1, dtype=torch.long)).flatten()


def torch_rotate(prev, idx):
    return prev[idx.clamp(min=0)] * (idx >= 0).float()[:, None, None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--workload", default="base")
    args = ap.parse_args()
    name = args.workload
    w = S.WORKLOADS[name]
    out = []
    mlvl, bq, kw = S.make_transformer_inputs(name, seed=0, temporal=True, device=DEV)
    g = torch.Generator().manual_seed(0)
    ce, le = torch.randn(6, 256, generator=g).to(DEV), torch.randn(len(mlvl), 256, generator=g).to(DEV)
    nbytes = 2 * sum(f.numel() for f in mlvl) * 4
    t_hip = timeit(lambda: ops.flatten_feats(mlvl, ce, le), args.iters)
    t_ref = timeit(lambda: torch_flatten(mlvl, ce, le).contiguous(), args.iters)
    out.append(dict(op="flatten_feats", hip_us=t_hip, torch_us=t_ref, alg_MB=nbytes / 1e6,
                    hip_GBs=nbytes / t_hip / 1e3))
    prev = kw["prev_bev"].permute(1, 0, 2).contiguous()
    idx = rotate_index(w["bev_h"], w["bev_w"], 4.0, [w["bev_w"] // 2, w["bev_h"] // 2]).to(DEV)
    nb = 2 * prev.numel() * 4
    t_hip = timeit(lambda: ops.rotate_bev(prev, [4.0], [w["bev_w"] // 2, w["bev_h"] // 2], w["bev_h"], w["bev_w"]), args.iters)
    t_ref = timeit(lambda: torch_rotate(prev, idx), args.iters)
    out.append(dict(op="rotate_bev", hip_us=t_hip, torch_us=t_ref, alg_MB=nb / 1e6, hip_GBs=nb / t_hip / 1e3))

    torch.manual_seed(0)
    t = bevformer_amd.build_transformer(S.transformer_cfg(name)).eval()
    t.init_weights()
    sd = t.state_dict()
    enc = S.trained_like_({k[8:]: v.clone() for k, v in sd.items() if k.startswith("encoder.")}, seed=3)
    t.encoder.load_state_dict(enc)
    t = t.to(DEV)
    with torch.no_grad():
        t_all = timeit(lambda: t.get_bev_features(mlvl, bq, **kw), args.iters)
        q, f, ekw = S.make_inputs(name, seed=0, temporal=True, device=DEV)
        t_enc = timeit(lambda: t.encoder(q, f, f, **ekw), args.iters)
    out.append(dict(op="get_bev_features (eager)", us=t_all, encoder_only_us=t_enc))
    for r in out:
        print(json.dumps(r))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/pbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
