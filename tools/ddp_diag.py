"""Forensics for tests/test_ddp_gpu.py (GPU box): repeats the two-rank scenario until a pass of the DDP-wrapped encoder
disagrees with a reference pass, then prints where (forward outputs, which gradient tensors, which elements).
    python tools/ddp_diag.py [--tries 6]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, name, tries):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from helpers import build_pair
    from bevformer_amd import synthetic as S
    from test_ddp_gpu import _grads
    w = S.WORKLOADS[name]
    Q = w["bev_h"] * w["bev_w"]
    enc, _ = build_pair(name, device=dev)
    for p in enc.parameters():
        p.requires_grad_(True)
    q, f, kw = S.make_inputs(name, seed=10 + rank, temporal=True, device=dev)
    gout = torch.randn(1, Q, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(20 + rank)) * 1e-2
    if os.environ.get("DDP_DIAG_BF16") == "1":      # BASELINE configs[2]'s arithmetic: bf16 value storage, bf16 GEMM operands
        from bevformer_amd import ops as _ops
        _ops.set_value_storage(torch.bfloat16)
        _ops.set_gemm_mode("bf16")
    ddp = DDP(enc, device_ids=[0], broadcast_buffers=False)
    mode = os.environ.get("DDP_DIAG_MODE", "ddp")
    glitches = 0
    for t in range(tries):
        # one DDP pass (all-reduce included), then direct passes of the same module: pass A right after it, pass B after A
        ddp.zero_grad(set_to_none=True)
        if mode == "ddp":
            out = ddp(q, f, f, **kw)
        else:
            out = ddp.module(q, f, f, **kw)
        out.backward(gout)
        if mode == "ddp_sync":
            torch.cuda.synchronize()
        if mode == "sleep":             # no DDP pass in front: an idle GPU for 50 ms, as a gloo all-reduce leaves it
            import time
            torch.cuda.synchronize()
            time.sleep(0.05)
        if mode == "alloc":             # no DDP pass: allocator churn in front of pass A (other blocks than pass B / C will get)
            junk = [torch.randn(n, device=dev) for n in (1 << 20, 3 << 18, 1 << 16, 5 << 14)]
            del junk
        recs = {}

        def tapped(name):
            """One direct pass; with DDP_DIAG_TAP=1 every operand / result of every fused sampling backward is kept."""
            if os.environ.get("DDP_DIAG_TAP") != "1":
                return _grads(ddp.module, q, f, kw, gout)
            from bevformer_amd import ops
            rec = recs[name] = []
            ops._DEBUG["tap"] = lambda tag, d: rec.append((tag, {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}))
            try:
                return _grads(ddp.module, q, f, kw, gout)
            finally:
                ops._DEBUG["tap"] = None
        oA, gA = tapped("A")
        oB, gB = tapped("B")
        oC, gC = tapped("C")
        def worst(a, b):
            floor = 1e-2 * max(v.norm().item() for v in b.values())
            return max((((a[k] - b[k]).norm() / max(b[k].norm().item(), floor)).item(), k) for k in b)
        wAB, wBC, wAC = worst(gA, gB), worst(gB, gC), worst(gA, gC)
        msg = f"rank {rank} try {t} mode {mode}: A-B {wAB[0]:.1e} B-C {wBC[0]:.1e} A-C {wAC[0]:.1e} fwd A-B {(oA - oB).abs().max().item():.1e}"
        if max(wAB[0], wBC[0], wAC[0]) > 2e-4:
            odd, ok1 = (gA, gB) if wBC[0] < 2e-4 else ((gB, gC) if wAC[0] < 2e-4 else (gC, gA))
            k = worst(odd, ok1)[1]
            d = (odd[k] - ok1[k]).abs()
            thr = 1e-3 * ok1[k].abs().max()
            bad = (d > thr).nonzero()
            msg += f"\n   odd tensor {k} shape {tuple(d.shape)}: {bad.shape[0]} of {d.numel()} elements off; first {bad[:5].tolist()} last {bad[-3:].tolist()}"
            if d.dim() == 2:
                rows = (d > thr).any(1).nonzero().flatten()
                cols = (d > thr).any(0).nonzero().flatten()
                msg += f"\n   rows {rows[:8].tolist()}..{rows[-3:].tolist()} ({rows.numel()}), cols {cols[:8].tolist()}..{cols[-3:].tolist()} ({cols.numel()})"
            i0 = tuple(bad[0].tolist())
            msg += f"\n   element {i0}: odd {odd[k][i0].item():+.6e} ok {ok1[k][i0].item():+.6e} (odd - ok = {(odd[k][i0] - ok1[k][i0]).item():+.3e}, max |grad| {ok1[k].abs().max().item():.3e})"
            others = sorted((((odd[n] - ok1[n]).norm() / (ok1[n].norm() + 1e-30)).item(), n) for n in ok1)[-5:]
            msg += "\n   plain rel L2 top: " + ", ".join(f"{n.replace('layers.', 'L')}: {e:.1e}" for e, n in reversed(others))
            if recs:
                names = ("A", "B") if wBC[0] < 2e-4 else (("B", "C") if wAC[0] < 2e-4 else ("C", "A"))
                for i, ((tag, do), (_, dk)) in enumerate(zip(recs[names[0]], recs[names[1]])):
                    n = int(do["nrows"].item()) if do["nrows"] is not None else None
                    for key in ("proj", "ref", "loc", "attn", "g", "gl", "ga", "gproj"):
                        a, b = do[key], dk[key]
                        if key in ("loc", "attn", "gl", "ga") and n is not None:
                            a, b = a[:n], b[:n]
                        a, b = a.float(), b.float()
                        exact = key in ("proj", "ref", "loc", "attn")
                        dd = (a - b).abs()
                        lim = 0.0 if exact else 1e-4 * b.abs().max().item()
                        nbad = int((dd > lim).sum().item())
                        if nbad:
                            idx = (dd > lim).nonzero()[:3].tolist()
                            vals = [(a[tuple(j)].item(), b[tuple(j)].item()) for j in idx]
                            msg += f"\n   call {i} {tag} {key} {tuple(a.shape)}: {nbad} elements differ (odd pass {names[0]} vs {names[1]}), at {idx} values {vals}"
                            if key == "gl" and nbad <= 8:
                                for j in (dd > lim).nonzero().tolist():
                                    r, m, l, pnt, c = j
                                    lx, ly = do["loc"][r, m, l, pnt].tolist()
                                    same = [bool(torch.equal(do[k2][r, m, l, pnt], dk[k2][r, m, l, pnt])) for k2 in ("loc", "attn", "ga")]
                                    gx_pair = (do["gl"][r, m, l, pnt, 0].item(), dk["gl"][r, m, l, pnt, 0].item())
                                    msg += (f"\n      (row {r}, head {m}, level {l}, point {pnt}, {'xy'[c]}): loc ({lx!r}, {ly!r}); loc / attn / grad_attn "
                                            f"bit-equal {same}; grad_loc.x odd {gx_pair[0]!r} ok {gx_pair[1]!r}; whole gl row of this (row, head) differs in "
                                            f"{int((do['gl'][r, m] != dk['gl'][r, m]).sum())} of {do['gl'][r, m].numel()} elements (bitwise)")
            glitches += 1
            print(msg, flush=True)
    print(f"rank {rank} mode {mode}: {glitches} of {tries} tries with a disagreeing pass", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    tries = int(sys.argv[sys.argv.index("--tries") + 1]) if "--tries" in sys.argv else 6
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, os.environ.get("DDP_DIAG_WORKLOAD", "micro4"), tries), nprocs=2, join=True)
