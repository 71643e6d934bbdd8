"""The round-5 scenario without DDP and without gloo: ONE process runs training passes of the encoder (forward +
backward, the same inputs every pass) and compares every parameter gradient of a pass bitwise with the first pass; a
contender of a chosen KIND shares the GPU.  Says what the second process has to be doing for the wrong grad_loc_y to appear.

    BEVMSDA_LIBRARY=.../libbevmsda_slp.so python tools/probes/pk_repro/flow_hunt.py --contender self --passes 400

contenders: none | self (this loop in a second process) | matmul | tiny | idle   (hunt.py's)  | thread (this loop on a second
            thread + stream of the SAME process)
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contender", default="self", choices=["none", "self", "matmul", "tiny", "idle", "thread"])
    ap.add_argument("--role", default="hunter")
    ap.add_argument("--passes", type=int, default=4000)
    ap.add_argument("--workload", default="micro4")
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--sync-every-pass", action="store_true", help="torch.cuda.synchronize() between forward and backward too")
    args = ap.parse_args()
    from helpers import build_pair
    from bevformer_amd import synthetic as S
    dev = torch.device("cuda", 0)
    w = S.WORKLOADS[args.workload]
    Q = w["bev_h"] * w["bev_w"]

    def make(seed):
        enc, _ = build_pair(args.workload, device=dev)
        for p in enc.parameters():
            p.requires_grad_(True)
        q, f, kw = S.make_inputs(args.workload, seed=seed, temporal=True, device=dev)
        gout = torch.randn(1, Q, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 10)) * 1e-2
        return enc, q, f, kw, gout

    def one_pass(enc, q, f, kw, gout):
        enc.zero_grad(set_to_none=True)
        out = enc(q, f, f, **kw)
        if args.sync_every_pass:
            torch.cuda.synchronize()
        out.backward(gout)
        return out.detach(), {k: p.grad.detach().clone() for k, p in enc.named_parameters()}

    model = make(args.seed)
    if args.role == "contender":
        while True:
            one_pass(*model)
            torch.cuda.synchronize()
    child, stop = None, False
    if args.contender in ("matmul", "tiny", "idle"):
        child = subprocess.Popen([sys.executable, os.path.join(HERE, "hunt.py"), "--role", "contender", "--contender",
                                  args.contender], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    elif args.contender == "self":
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--role", "contender", "--seed", str(args.seed + 1),
                                  "--workload", args.workload], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    elif args.contender == "thread":
        other = make(args.seed + 1)
        side = torch.cuda.Stream()

        def spin():
            with torch.cuda.stream(side):
                while not stop:
                    one_pass(*other)
                    side.synchronize()
        threading.Thread(target=spin, daemon=True).start()
    if child is not None:
        time.sleep(15.0)
    o0, g0 = one_pass(*model)
    bad, events, fwd_bad = 0, [], 0
    t0 = time.time()
    try:
        for i in range(args.passes):
            o, g = one_pass(*model)
            fwd_bad += int(not torch.equal(o, o0))
            # (grad_value goes through fp32 atomics: not bitwise repeatable; the criterion of tools/ddp_diag.py instead)
            floor = 1e-2 * max(v.norm().item() for v in g0.values())
            diff = [k for k in g0 if ((g[k] - g0[k]).norm() / max(g0[k].norm().item(), floor)).item() > 2e-4]
            if diff:
                bad += 1
                if len(events) < 12:
                    k = max(diff, key=lambda k: ((g[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)).item())
                    d = (g[k] - g0[k]).abs()
                    idx = (d > 1e-3 * g0[k].abs().max()).nonzero()
                    events.append(dict(pass_=i, n_tensors=len(diff), worst=k, elems=idx[:4].tolist(), n_elems=int(idx.shape[0])))
    finally:
        stop = True
        if child is not None:
            child.kill()
    print(json.dumps(dict(lib=os.path.basename(os.environ.get("BEVMSDA_LIBRARY", "default")), contender=args.contender,
                          workload=args.workload, passes=args.passes, seconds=round(time.time() - t0, 1), bad_passes=bad,
                          forward_outputs_differ=fwd_bad, events=events)), flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
