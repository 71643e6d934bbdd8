"""Run-to-run spread of the encoder's parameter gradients on the GPU box: the same encoder, the same inputs, N
forward + backward passes; per tensor the largest relative L2 distance from pass 0 (floored at 1 % of the largest gradient
norm).  The backward kernels' atomics make rounding-level spread (1e-7 .. 1e-6) normal; anything larger is a race or
a read of uninitialised memory.  Optional mode switches localise a finding.

    python tools/grad_determinism.py [--workload micro4] [--passes 6] [--modes default,fused_save=0,chain_backward=0,train_chain=0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bevformer_amd  # noqa: E402,F401
from bevformer_amd import ops  # noqa: E402
from bevformer_amd import synthetic as S  # noqa: E402
from helpers import build_pair  # noqa: E402

DEV = torch.device("cuda:0")


def run(enc, q, f, kw, gout):
    enc.zero_grad(set_to_none=True)
    out = enc(q, f, f, **kw)
    out.backward(gout)
    return {k: p.grad.detach().clone() for k, p in enc.named_parameters()}, out.detach().clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="micro4,tiny")
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--modes", default="default,fused_save=0,chain_backward=0,train_chain=0")
    ap.add_argument("--fresh", action="store_true", help="a freshly built encoder (same weights) for every pass")
    ap.add_argument("--steps", type=int, default=0, help="SGD steps (lr 1e-3) on the encoder before the passes; with --fresh the "
                    "fresh encoders load its state_dict (the comparison tests/test_ddp_gpu.py makes)")
    ap.add_argument("--seed", type=int, default=10)
    args = ap.parse_args()
    for name in args.workload.split(","):
        w = S.WORKLOADS[name]
        Q = w["bev_h"] * w["bev_w"]
        q, f, kw = S.make_inputs(name, seed=args.seed, temporal=True, device=DEV)
        gout = torch.randn(1, Q, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(args.seed + 10)) * 1e-2
        for mode in args.modes.split(","):
            over = {}
            if mode != "default":
                k, v = mode.split("=")
                over[k] = bool(int(v))
            with ops.using(**over):
                enc, sd = build_pair(name, device=DEV)
                for p in enc.parameters():
                    p.requires_grad_(True)
                if args.steps:
                    opt = torch.optim.SGD(enc.parameters(), lr=1e-3)
                    for _ in range(args.steps):
                        run(enc, q, f, kw, gout)
                        opt.step()
                    sd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
                base, out0 = run(enc, q, f, kw, gout)
                floor = 1e-2 * max(v.norm().item() for v in base.values())
                worst = {}
                outdiff = 0.0
                for i in range(1, args.passes):
                    if args.fresh:
                        enc, _ = build_pair(name, device=DEV)
                        if args.steps:
                            enc.load_state_dict(sd)
                        for p in enc.parameters():
                            p.requires_grad_(True)
                    g, out = run(enc, q, f, kw, gout)
                    outdiff = max(outdiff, (out - out0).abs().max().item())
                    for k in g:
                        e = ((g[k] - base[k]).norm() / max(base[k].norm().item(), floor)).item()
                        worst[k] = max(worst.get(k, 0.0), e)
                top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
                print(f"{name:7s} {mode:18s} fresh={args.fresh}  forward max |out - out0| {outdiff:.1e}   worst: " +
                      ", ".join(f"{k.replace('layers.', 'L').replace('attentions.', 'att').replace('deformable_attention.', 'da.')} {v:.1e}"
                                for k, v in top))


if __name__ == "__main__":
    main()
