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
        def agree(flag):
            """Both ranks repeat a pass if either saw a mismatch (the all-reduce inside it is collective)."""
            t = torch.tensor([1 if flag else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(t.item())

        for it in range(2):
            want = None
            for r in range(world):
                _, g = _grads(ref, *ins[r])
                want = g if want is None else {k: want[k] + g[k] for k in g}
            want = {k: v / world for k, v in want.items()}
            q, f, kw, gout = ins[rank]
            glitches = []
            for attempt in range(4):
                before = train_ops.stats()
                ddp.zero_grad(set_to_none=True)
                out = ddp(q, f, f, **kw)
                out.backward(gout)
                after = train_ops.stats()
                got = {}
                for k, p in ddp.module.named_parameters():
                    assert p.grad is not None, k
                    got[k] = p.grad
                err, where = _rel_errors(got, want)
                # (Round 5 found this scenario — a second process on the same GPU — computing, about once in fifty passes, ONE
                # wrong grad_loc_y in msda_gradloc_d32_kernel: packed fp32 math in that kernel, profiles/r5/r5_ddp_forensics.txt;
                # the sampling-backward unit is built without it since.  The loop that repeats a disagreeing pass on both
                # ranks stays as the detector: the test now demands that NO pass disagrees.)
                if not agree(err > 2e-4):
                    break
                glitches.append((attempt, err, where))
            report[f"iter{it}_worst_rel_l2"], report[f"iter{it}_worst_tensor"] = err, where
            report[f"iter{it}_glitched_passes"] = glitches
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
            # ... and from here on bit-equal: the two updates differ by the rounding of their mean gradients, and a weight
            # difference of 1e-9 is enough to move a sampling point of `tiny` across a pixel boundary (its grad_loc_y then takes
            # the neighbouring cell's slope: 3e-3 of the offset-bias gradient, seen twice at iter 1) — a property of bilinear
            # sampling, not of the reducer under test
            with torch.no_grad():
                for (_, a), (_, b) in zip(ddp.module.named_parameters(), ref.named_parameters()):
                    b.copy_(a)
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
        _, g_new = _grads(fresh, *ins[rank])
        stale = []
        for attempt in range(4):
            _, g_old = _grads(ddp.module, *ins[rank])
            err, where = _rel_errors(g_old, g_new)
            if err <= 2e-4:
                break
            stale.append((attempt, err, where))
        report["stale_cache_worst_rel_l2"], report["stale_cache_worst_tensor"] = err, where
        report["stale_cache_glitched_passes"] = stale
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
            assert not rep[f"iter{it}_glitched_passes"], (r, rep)
            assert rep[f"iter{it}_fast_path_seams"] > 0, "the training fast path did not run under DDP"
            assert rep[f"iter{it}_weights_equal_after_step"], (r, rep)
        assert rep["state_dict_round_trip_max_abs"] < 1e-5 and rep["state_dict_keys"], (r, rep)
        assert rep["stale_cache_worst_rel_l2"] < 2e-4 and not rep["stale_cache_glitched_passes"], (r, rep)
