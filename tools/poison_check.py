"""Does the training path read memory it never wrote?  Every float tensor that ``torch.empty`` / ``empty_like`` /
``new_empty`` hands out while the encoder runs is filled with NaN; a NaN in the output or in a gradient then marks a
read of an element no kernel stored (normally masked: the caching allocator hands a pass the same blocks as the pass
before, whose contents are the same numbers).      python tools/poison_check.py [--workload micro4,tiny,small4]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bevformer_amd import ops  # noqa: E402
from bevformer_amd import synthetic as S  # noqa: E402
from helpers import build_pair  # noqa: E402

DEV = torch.device("cuda:0")
_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
COUNT = {"n": 0}
VALUE = {"v": float("nan")}


INT = {"v": None}


def _poison(t):
    if t.is_floating_point() and t.is_cuda and t.numel():
        t.fill_(VALUE["v"])
        COUNT["n"] += 1
    elif INT["v"] is not None and t.is_cuda and t.numel() and t.dtype in (torch.int32, torch.int64, torch.uint8, torch.bool):
        t.fill_(INT["v"])          # an index table nobody filled: -1 reads as "absent", 0 as the wrong row
        COUNT["n"] += 1
    return t


def patch(on):
    if on:
        torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
        torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
        torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))
    else:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = _empty, _empty_like, _new_empty


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="micro4,tiny,small4")
    ap.add_argument("--modes", default="default,fused_save=0,chain_backward=0")
    ap.add_argument("--value", type=float, default=float("nan"), help="the poison (NaN; or e.g. 1e30 to see sizes)")
    ap.add_argument("--int-value", type=int, default=None, help="also poison integer / bool buffers with this value")
    args = ap.parse_args()
    VALUE["v"] = args.value
    INT["v"] = args.int_value
    for name in args.workload.split(","):
        w = S.WORKLOADS[name]
        Q = w["bev_h"] * w["bev_w"]
        for temporal in (True, False):
            for mode in args.modes.split(","):
                over = {}
                if mode != "default":
                    k, v = mode.split("=")
                    over[k] = bool(int(v))
                with ops.using(**over):
                    enc, _ = build_pair(name, device=DEV)
                    for p in enc.parameters():
                        p.requires_grad_(True)
                    q, f, kw = S.make_inputs(name, seed=10, temporal=temporal, device=DEV)
                    gout = torch.randn(1, Q, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(20)) * 1e-2
                    qg = q.clone().requires_grad_(True)
                    fg = f.clone().requires_grad_(True)
                    enc(qg, fg, fg, **kw).backward(gout)       # un-poisoned pass: planner, caches, images
                    clean = {k: p.grad.clone() for k, p in enc.named_parameters()}
                    enc.zero_grad(set_to_none=True)
                    qg.grad = fg.grad = None
                    COUNT["n"] = 0
                    patch(True)
                    try:
                        out = enc(qg, fg, fg, **kw)
                        out.backward(gout)
                    finally:
                        patch(False)
                    bad = [(k, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for k, p in enc.named_parameters()
                           if not torch.isfinite(p.grad).all()]
                    bad += [(k, int((~torch.isfinite(t)).sum()), t.numel()) for k, t in (("out", out), ("d bev_query", qg.grad), ("d feat", fg.grad))
                            if not torch.isfinite(t).all()]
                    diff = max(((p.grad - clean[k]).norm() / (clean[k].norm() + 1e-30)).item() for k, p in enc.named_parameters()
                               if torch.isfinite(p.grad).all()) if len(bad) < len(clean) else float("nan")
                    print(f"{name:7s} history={temporal!s:5s} {mode:18s} poisoned buffers {COUNT['n']:4d}  non-finite: "
                          f"{bad[:6] if bad else 'none'}{' ...' if len(bad) > 6 else ''}   worst finite rel diff vs clean pass {diff:.1e}", flush=True)


if __name__ == "__main__":
    main()
