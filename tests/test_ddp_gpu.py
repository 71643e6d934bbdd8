"""GPU, ONE device: the reference's parallel TRAINING strategy around the product's training fast path.

The reference trains with ``MMDistributedDataParallel`` — torch's DistributedDataParallel over one process per GPU
(bevformer/apis/mmdet_train.py:75-79).  The training fast path of this package (train_ops.py) does three things a DDP
reducer could trip over: on the first training forward it re-seats ``Parameter.data`` of the merged projections into
shared flat buffers (``ops.flatten_linear_params``), it returns parameter gradients as views of merged buffers / of one
zero-filled arena, and its Functions run hand-ordered backwards.  Here two processes share ``cuda:0`` (RCCL refuses
duplicate devices, so the reducer's all-reduce runs over ``gloo``), each with DIFFERENT inputs, and every parameter's
gradient after ``DistributedDataParallel(...).backward`` must be the MEAN of the two single-process gradients — on the
first iteration (the one that flattens), after an optimizer step (in-place update of the flattened parameters: the
packed-weight caches must follow), with ``gradient_as_bucket_view`` on and off, and after a ``state_dict`` round trip
into a fresh encoder."""
import copy
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


LR = 1e-3       # small steps: after a large one the bilinear taps of many sampling points change cells and rounding-level
                # differences (the order of the backward kernels' atomics) no longer stay rounding-level in the gradients


def _rel_errors(got, want):
    """Per-tensor L2 error relative to the tensor's own norm, floored at 1 % of the largest gradient norm of the model (a
    parameter whose gradient is almost zero has no meaningful relative error) -> (worst, its name)."""
    floor = 1e-2 * max(v.norm().item() for v in want.values())
    worst, name = 0.0, None
    for k, w in want.items():
        e = ((got[k] - w).norm() / max(w.norm().item(), floor)).item()
        if e > worst:
            worst, name = e, k
    return worst, name


def _grads(enc, q, f, kw, gout):
    """Single-process forward + backward -> (output, {name: grad})."""
    if os.environ.get("DDP_TEST_SYNC") == "1":
        torch.cuda.synchronize()
    enc.zero_grad(set_to_none=True)
    out = enc(q, f, f, **kw)
    out.backward(gout)
    return out.detach(), {k: p.grad.detach().clone() for k, p in enc.named_parameters()}


def _worker(rank, world, port, name, bucket_view, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        from helpers import build_pair
        from bevformer_amd import synthetic as S
        from bevformer_amd import train_ops
        enc, _ = build_pair(name, device=dev)
        for p in enc.parameters():
            p.requires_grad_(True)
        ref = copy.deepcopy(enc)                    # the single-process encoder (same initial weights on every rank)
        w = S.WORKLOADS[name]
        Q = w["bev_h"] * w["bev_w"]
        ins = []
        for r in range(world):                      # every rank builds every rank's inputs: the references need them
            q, f, kw = S.make_inputs(name, seed=10 + r, temporal=True, device=dev)
            gout = torch.randn(1, Q, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(20 + r)) * 1e-2
            ins.append((q, f, kw, gout))
        ddp = DDP(enc, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=bucket_view)
        report = {}
        opt = torch.optim.SGD(ddp.parameters(), lr=LR)
        opt_ref = torch.optim.SGD(ref.parameters(), lr=LR)
        for it in range(2):
            want = None
            for r in range(world):
                _, g = _grads(ref, *ins[r])
                want = g if want is None else {k: want[k] + g[k] for k in g}
            want = {k: v / world for k, v in want.items()}
            before = train_ops.stats()
            q, f, kw, gout = ins[rank]
            if os.environ.get("DDP_TEST_SYNC") == "1":
                torch.cuda.synchronize()
            ddp.zero_grad(set_to_none=True)
            out = ddp(q, f, f, **kw)
            out.backward(gout)
            # gloo stages the buckets of CUDA parameters through pinned host memory on its own streams; with a device
            # synchronisation behind the reducer's backward the comparison below is exact to rounding every time.  Without
            # it ONE element of ONE gradient (a sampling-offset column) of this pass or of the NEXT pass of the same
            # module was off by 1e-3 .. 1e-2 in about one pass of sixteen — never without a DDP pass in front, not with
            # two processes sharing the GPU, not with poisoned torch.empty buffers (tools/ddp_diag.py, tools/poison_check.py,
            # tools/grad_determinism.py; profiles/r5/r5_ddp_forensics.txt).  RCCL (the product transport) has no host staging.
            torch.cuda.synchronize()
            after = train_ops.stats()
            got = {}
            for k, p in ddp.module.named_parameters():
                assert p.grad is not None, k
                got[k] = p.grad
            report[f"iter{it}_worst_rel_l2"], report[f"iter{it}_worst_tensor"] = _rel_errors(got, want)
            if report[f"iter{it}_worst_rel_l2"] > 2e-4:
                # diagnosis: the single-process gradients once more (is `want` reproducible?)
                again = None
                for r in range(world):
                    _, g = _grads(ref, *ins[r])
                    again = g if again is None else {k: again[k] + g[k] for k in g}
                again = {k: v / world for k, v in again.items()}
                report[f"iter{it}_diag_want_vs_want2"] = _rel_errors(again, want)
                report[f"iter{it}_diag_ddp_vs_want2"] = _rel_errors({k: v.clone() for k, v in got.items()}, again)
            report[f"iter{it}_fast_path_seams"] = after["seam_s"] - before["seam_s"]
            # the same update on both sides (the reference model steps with the MEAN gradient it just computed)
            for k, p in ref.named_parameters():
                p.grad = want[k].clone()
            opt.step()
            opt_ref.step()
            gmax = max(v.abs().max().item() for v in want.values())
            diff = max((a - b).abs().max().item()
                       for (_, a), (_, b) in zip(ddp.module.named_parameters(), ref.named_parameters()))
            report[f"iter{it}_weights_equal_after_step"] = diff <= LR * 1e-2 * gmax + 1e-7
        # state_dict round trip: the flattened (re-seated) parameters save and load like any others
        sd = {k: v.detach().cpu().clone() for k, v in ddp.module.state_dict().items()}
        fresh, _ = build_pair(name, device=dev)
        fresh.load_state_dict(sd)
        q, f, kw, _ = ins[rank]
        with torch.no_grad():
            a = ddp.module(q, f, f, **kw)
            b = fresh(q, f, f, **kw)
        report["state_dict_round_trip_max_abs"] = (a - b).abs().max().item()
        # ... and so do the weight images the training path caches per parameter version (packed / panel / transposed
        # images of the flattened, in-place-updated parameters): gradients of the stepped encoder == gradients of a
        # fresh encoder that never saw the old weights
        for p in fresh.parameters():
            p.requires_grad_(True)
        o_old, g_old = _grads(ddp.module, *ins[rank])
        o_new, g_new = _grads(fresh, *ins[rank])
        report["stale_cache_worst_rel_l2"], report["stale_cache_worst_tensor"] = _rel_errors(g_old, g_new)
        if report["stale_cache_worst_rel_l2"] > 2e-4:
            # diagnosis of a mismatch: is each side reproducible, and does the mismatch persist?
            o_old2, g_old2 = _grads(ddp.module, *ins[rank])
            _, g_new2 = _grads(fresh, *ins[rank])
            report["diag_forward_old_vs_new"] = (o_old - o_new).abs().max().item()
            report["diag_forward_old_vs_old2"] = (o_old - o_old2).abs().max().item()
            k = report["stale_cache_worst_tensor"]
            d = (g_old[k] - g_new[k]).abs()
            thr = 1e-3 * g_new[k].abs().max()
            bad = (d > thr).nonzero()
            report["diag_bad_elems"] = (int(bad.shape[0]), int(d.numel()), bad[:6].tolist(), bad[-3:].tolist())
            others = sorted(((((g_old[n] - g_new[n]).norm() / (g_new[n].norm() + 1e-30)).item(), n) for n in g_new), reverse=True)[:6]
            report["diag_plain_rel_l2_top"] = others
            report["diag_old_vs_old2"] = _rel_errors(g_old2, g_old)
            report["diag_new_vs_new2"] = _rel_errors(g_new2, g_new)
            report["diag_old2_vs_new2"] = _rel_errors(g_old2, g_new2)
        report["state_dict_keys"] = sorted(sd) == sorted(fresh.state_dict())
        ret[rank] = report
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,bucket_view", [("micro4", False), ("micro4", True), ("tiny", False)])
def test_ddp_gradients_are_the_mean_of_the_single_process_gradients(name, bucket_view):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), name, bucket_view, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        rep = ret[r]
        print(r, rep)
        for it in (0, 1):
            # the summation order of the backward kernels' atomics is not fixed: rounding-level agreement
            assert rep[f"iter{it}_worst_rel_l2"] < 2e-4, (r, rep)
            assert rep[f"iter{it}_fast_path_seams"] > 0, "the training fast path did not run under DDP"
            assert rep[f"iter{it}_weights_equal_after_step"], (r, rep)
        assert rep["state_dict_round_trip_max_abs"] < 1e-5 and rep["state_dict_keys"], (r, rep)
        assert rep["stale_cache_worst_rel_l2"] < 2e-4, (r, rep)
