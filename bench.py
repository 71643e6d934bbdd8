"""BEV-encoder throughput bench (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload base] [--dtype fp32]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward pass of the whole BEV encoder (all layers: TSA -> LN ->
SCA -> LN -> FFN -> LN, including the per-frame geometry plan lookup) over one
synthetic frame of the workload (default ``base`` = bevformer_base: 200x200 BEV
queries, 6 cameras, 4 feature levels, 6 layers) with a history BEV
(``prev_bev``) present, inputs resident in HBM.  ``value`` = BEV queries / s for
the whole job.  With N > 1 the ONE frame is tiled over the N GPUs by BEV rows
(strong scaling) and reassembled with an RCCL all-gather inside the timed step.

Prints one JSON line on rank 0 with the extra objects ``roofline`` (dominant
hand-written kernel: the SCA deformable-sampling forward, timed live with HIP
events on its launch stream) and ``cpu_baseline`` (the oracle's pure-PyTorch
CPU port of the same encoder, timed on the host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="base")
    ap.add_argument("--dtype", default="fp32", choices=["fp32"])
    ap.add_argument("--gemm", default=None, choices=["split", "bf16", "native"],
                    help="how the Linear layers run (bevformer_amd.ops.set_gemm_mode); default: the "
                         "package default / BEVMSDA_GEMM")
    ap.add_argument("--value-storage", default="fp32", choices=["fp32", "bf16"],
                    help="storage of the projected value tensors (bf16: written by the projection "
                         "kernel, sampled by the 16-byte-lane bf16 kernel; arithmetic stays fp32)")
    ap.add_argument("--sca-lds", default=None, choices=["on", "off"],
                    help="SCA sampling kernel with the coarsest level staged in LDS (default: package default)")
    ap.add_argument("--queue", type=int, default=0,
                    help="N > 0: a step is N consecutive frames through PerceptionTransformer.get_bev_features "
                         "(ego-motion shift, prev-BEV rotation, can-bus MLP, flatten + embeddings, encoder), each "
                         "frame's BEV being the next frame's history (BASELINE configs[4] style; eager launches)")
    ap.add_argument("--backward", action="store_true",
                    help="time forward + backward of the encoder (autograd path: unfused operator with its "
                         "backward kernels, hipBLASLt fp32 GEMMs; eager launches; BASELINE configs[2] style)")
    ap.add_argument("--first-frame", action="store_true", help="no history BEV (prev_bev=None)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a captured HIP graph in the timed region "
                         "(auto = on, falling back to eager launches if the capture fails; the "
                         "per-kernel HIP-event timings come from an eager pass right before)")
    ap.add_argument("--force-tiling", action="store_true",
                    help="run the multi-GPU schedule (row blocks + RCCL all-gather) even on 1 rank")
    ap.add_argument("--row-order", default=None, choices=["raster", "image"],
                    help="order of the ragged SCA rows inside a camera (default: the encoder's)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch of the roofline kernel (optional)")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of the sampling-kernel launches inside the timed
    region, recorded on the stream the kernel is launched on."""

    def __init__(self):
        self.events = []   # (tag, start, end, alg_bytes)
        self.enabled = False

    def __call__(self, tag, alg_bytes):
        timer = self

        class _Ctx:
            def __enter__(self_inner):
                if timer.enabled:
                    self_inner.s = torch.cuda.Event(enable_timing=True)
                    self_inner.e = torch.cuda.Event(enable_timing=True)
                    self_inner.s.record()
                return self_inner

            def __exit__(self_inner, *exc):
                if timer.enabled:
                    self_inner.e.record()
                    timer.events.append((tag, self_inner.s, self_inner.e, alg_bytes))
                return False
        return _Ctx()

    def gemm(self, tag, flops, nbytes):
        """Same bracket for the projection GEMM launches (ops.set_gemm_timer)."""
        return self("gemm:" + tag, (flops, nbytes))

    def summary(self):
        agg = {}
        for tag, s, e, b in self.events:
            if tag.startswith("gemm:"):
                continue
            a = agg.setdefault(tag, [0.0, 0, 0])
            a[0] += s.elapsed_time(e) * 1e-3
            a[1] += 1
            a[2] += b
        return {t: dict(avg_us=a[0] / a[1] * 1e6, launches=a[1], alg_bytes=a[2] / a[1],
                        GBs=a[2] / a[0] / 1e9) for t, a in agg.items()}

    def gemm_summary(self):
        agg = {}
        for tag, s, e, b in self.events:
            if not tag.startswith("gemm:"):
                continue
            a = agg.setdefault(tag[5:], [0.0, 0, 0.0, 0.0])
            a[0] += s.elapsed_time(e) * 1e-3
            a[1] += 1
            a[2] += b[0]
            a[3] += b[1]
        per = {t: dict(avg_us=a[0] / a[1] * 1e6, launches=a[1], TFLOPs=a[2] / a[0] / 1e12,
                       alg_GBs=a[3] / a[0] / 1e9) for t, a in agg.items()}
        if not agg:
            return None
        tot_t = sum(a[0] for a in agg.values())
        return dict(per_tag=per, total_us_per_step=None, TFLOPs=sum(a[2] for a in agg.values()) / tot_t / 1e12,
                    alg_GBs=sum(a[3] for a in agg.values()) / tot_t / 1e9, seconds=tot_t,
                    launches=sum(a[1] for a in agg.values()))


def _pick_cpu_threads(cores):
    """Thread count for the CPU leg: the pure-PyTorch fallback is made of many small
    ops and gets SLOWER with every hardware thread of a 2-socket host (measured: 256
    threads -> 135 s per base frame vs ~14 s on 8), so probe a tiny frame at a few
    counts and keep the fastest.  Returns (threads, {threads: seconds})."""
    import bevformer_amd
    from bevformer_amd import synthetic as S
    from oracle import bevformer_cpu as O
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg("tiny")).eval()
    sd = S.trained_like_({k: v.clone() for k, v in enc.state_dict().items()}, seed=3)
    q, f, kw = S.make_inputs("tiny", seed=0, temporal=True)
    cands = sorted({c for c in (8, 16, 32, 64, 128) if c <= cores} | {min(cores, 8)})
    seen = {}
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)      # warm the pool
            t0 = time.perf_counter()
            O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
            seen[c] = time.perf_counter() - t0
    best = min(seen, key=seen.get)
    return best, seen


def cpu_baseline(workload, sd, first_frame):
    """The oracle's CPU port of the encoder on the host cores (bounded sample:
    ONE frame of the same workload; fp32, no_grad; thread count picked by a probe)."""
    from bevformer_amd import synthetic as S
    from oracle import bevformer_cpu as O
    cores = os.cpu_count() or 1
    threads, probe = _pick_cpu_threads(cores)
    torch.set_num_threads(threads)
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    w = S.WORKLOADS[workload]
    with torch.no_grad():
        q, f, kw = S.make_inputs(workload, seed=0, temporal=not first_frame)
        t0 = time.perf_counter()
        O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
        dt = time.perf_counter() - t0
    Q = w["bev_h"] * w["bev_w"]
    return dict(value=Q / dt, unit="BEV queries/s", cores=threads, kind="port",
                seconds=dt, host_cpus=cores,
                thread_probe_tiny_frame_s={str(k): round(v, 3) for k, v in probe.items()},
                sample=f"1 frame of {workload} ({w['layers']} layers, {Q} queries, "
                       f"{'no ' if first_frame else ''}history BEV) through oracle/bevformer_cpu.py "
                       "(pure-PyTorch CPU fallback path of the reference, fp32, no_grad)")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    elif args.force_tiling:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    import bevformer_amd
    from bevformer_amd import bev_tiling, ops
    from bevformer_amd import synthetic as S

    torch.manual_seed(0)
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg(args.workload)).eval()
    sd = S.trained_like_({k: v.clone() for k, v in enc.state_dict().items()}, seed=3)
    enc.load_state_dict(sd)
    enc = enc.to(dev)
    if args.row_order:
        enc.sca_row_order = args.row_order
    if world > 1 or args.force_tiling:
        bev_tiling.enable_bev_tiling(enc)
    q, f, kw = S.make_inputs(args.workload, seed=0, temporal=not args.first_frame, device=dev)
    w = S.WORKLOADS[args.workload]
    Q = w["bev_h"] * w["bev_w"]

    if args.gemm:
        ops.set_gemm_mode(args.gemm)
    if args.sca_lds:
        ops.set_sca_lds_level(args.sca_lds == "on")
    if args.value_storage == "bf16":
        ops.set_value_storage(torch.bfloat16)
    timer = KernelTimer()
    ops.set_kernel_timer(timer)
    ops.set_gemm_timer(timer.gemm)

    if args.queue > 0:
        tr = bevformer_amd.build_transformer(S.transformer_cfg(args.workload)).eval()
        tr.init_weights()
        tr.encoder = enc                       # the encoder above (trained-like weights, tiling if enabled)
        tr = tr.to(dev)
        mlvl, bq, tkw = S.make_transformer_inputs(args.workload, seed=0, temporal=False, device=dev)
        tkw.pop("prev_bev")
    if args.backward:
        g_out = torch.randn(1, Q, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        qg, fg = q.clone().requires_grad_(True), f.clone().requires_grad_(True)

    def step():
        if args.queue > 0:      # frame i's BEV is frame i+1's history; frame 0 has none
            with torch.no_grad():
                prev = None
                for _ in range(args.queue):
                    prev = tr.get_bev_features(mlvl, bq, prev_bev=prev, **tkw)
            return prev
        if args.backward:       # fwd + bwd w.r.t. parameters, BEV queries and camera features
            enc.zero_grad(set_to_none=True)
            qg.grad = fg.grad = None
            out = enc(qg, fg, fg, **kw)
            out.backward(g_out)
            return out.detach()
        with torch.no_grad():
            return enc(q, f, f, **kw)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or args.force_tiling:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    # Timed region: replays of ONE captured HIP graph of the whole step (~100 launches per
    # step; with N > 1 the collective is captured too), per-kernel durations from HIP events
    # around every sampling / GEMM launch of two eager steps right before; or (--graph off, or
    # a failed capture) eager launches with the events recorded inside the timed region.
    graph = None
    use_graph = args.graph in ("on", "auto") and not args.backward and args.queue == 0
    graph_note = "eager"
    if use_graph:
        timer.enabled = True            # kernel durations from an eager pass (events cannot
        for _ in range(2):              # bracket nodes inside a graph replay)
            step()
        fence()
        timer.enabled = False
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                  # allocator warm-up on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            fence()
            graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread keeps polling events of earlier eager
            # collectives; in the default "global" mode its hipEventQuery during our capture
            # aborts the process ("operation not permitted when stream is capturing")
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                g_out = step()
            graph.replay()
            fence()
            graph_note = "hip graph replay"
        except Exception as e:          # noqa: BLE001 — any capture problem: measure eagerly
            graph = None
            graph_note = f"eager (graph capture failed: {type(e).__name__}: {str(e)[:120]})"
            torch.cuda.synchronize()
    if graph is None:
        timer.enabled = not timer.events
    fence()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps):
            graph.replay()
        out = g_out
    else:
        for _ in range(args.steps):
            out = step()
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        ks = timer.summary()
        gs = timer.gemm_summary()
        if gs is not None:
            steps_timed = max(1, ks.get("sca_fwd", {"launches": w["layers"]})["launches"] // w["layers"])
            gs["total_us_per_step"] = gs.pop("seconds") / steps_timed * 1e6
            # fractions of the two MFMA peaks (MI355X_MICROARCH.md): the split kernel issues 3 bf16
            # products per algorithmic product -> its matrix-core utilisation is 3 * TFLOPs / bf16 peak
            gs["frac_of_f32_mfma_peak_157"] = gs["TFLOPs"] / 157.3
            if ops.gemm_mode() != "native":
                gs["frac_of_bf16_mfma_peak_2500"] = gs["TFLOPs"] * (3 if ops.gemm_mode() == "split" else 1) / 2500.0
        dom = ks.get("sca_fwd") or next(iter(ks.values()))
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                traffic = json.load(open(args.traffic_json)).get(args.workload, {}).get("sca_fwd")
            except Exception:
                traffic = None
        # what the timed steps used: the encoder picks the row order by grad mode
        with torch.set_grad_enabled(args.backward):
            row_order_used = enc.row_order()
            rows_per_frame = int(sum(enc.frame_plan(w['bev_h'], w['bev_w'], 1, kw['img_metas'], dev, torch.float32).hits))
        line = {
            "metric": "BEV-encoder queries/sec (200x200 BEV, 6 cams, 4 lvls)" if args.workload == "base"
            else f"BEV-encoder queries/sec ({args.workload})",
            "value": Q * max(1, args.queue) * args.steps / dt, "unit": "BEV queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # arithmetic type of the path: fp32 accumulation / sampling / softmax / LayerNorm always;
            # "bf16" when the GEMM operands are rounded to bf16 (--gemm bf16)
            "dtype": "bf16" if ops.gemm_mode() == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"bevformer_{args.workload} BEV encoder {'forward + backward' if args.backward else 'forward'}, "
                                   f"{args.queue if args.queue else 1} frame{'s' if args.queue > 1 else ''}/step"
                                   f"{' through get_bev_features with a rolling history BEV' if args.queue else ''}, "
                                   f"{w['bev_h']}x{w['bev_w']} queries, 6 cams, {len(w['shapes'])} levels, "
                                   f"{w['layers']} layers, " + ("frame 0 without history, frames 1.. with the previous frame's BEV"
                                                           if args.queue else
                                                           ('first frame (no history)' if args.first_frame else 'with history BEV')),
                       "sca_row_order": row_order_used,
                       "sca_coarse_level_from_lds": bool(ops._FUSED["lds_level"]),
                       "value_storage": args.value_storage,
                       "gemm": {"split": "hand-written MFMA kernel, fp32 operands split into 2 bf16 terms, "
                                         "3 bf16 MFMA products per fp32 product, fp32 accumulate",
                                "bf16": "hand-written MFMA kernel, operands rounded to bf16, fp32 accumulate",
                                "native": "hipBLASLt fp32 (torch.nn.functional.linear)"}[ops.gemm_mode()],
                       "global_batch": 1, "parallelism": f"bev-row-tiles x{world}" if world > 1 else "single GPU",
                       "sca_rows_per_frame": rows_per_frame},
            "roofline": {"kernel": "msda_fwd (SCA sampling, ragged rows)", "bound": "hbm",
                         "achieved": dom["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_us": dom["avg_us"], "alg_bytes": dom["alg_bytes"],
                         "launches_timed": dom["launches"],
                         "timing": "HIP events on the launch stream, " +
                                   ("every launch of the timed region" if graph is None and use_graph is False
                                    else "eager pass right before the timed region")},
            "launch_mode": graph_note,
            "kernels": ks,
            "gemms": gs,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, sd, args.first_frame)
            line["vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    else:
        line = None
    if world > 1 or args.force_tiling:
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes a version banner through C stdio, flushed at exit: push it out first so
        # that the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
