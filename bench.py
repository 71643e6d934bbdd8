"""BEV-encoder throughput bench (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload base]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward pass of the whole BEV encoder (per-frame geometry: camera
projection, visibility, ragged row lists — then all layers: TSA -> LN -> SCA -> LN ->
FFN -> LN) over one synthetic frame of the workload (default ``base`` =
bevformer_base: 200x200 BEV queries, 6 cameras, 4 feature levels, 6 layers) with a
history BEV (``prev_bev``) present, inputs resident in HBM.  Every step gets NEW camera
matrices (a seeded ego-pose jitter of the synthetic rig, as nuScenes rebuilds
``lidar2img`` per sample), so the frame plan is rebuilt inside the timed region.
``value`` = BEV queries / s for the whole job.  With N > 1 the ONE frame is tiled over the
N GPUs by BEV rows (strong scaling) and reassembled with an RCCL all-gather inside the
timed step.

Rank 0 prints TWO JSON lines: first ``{"bench_detail": {...}}`` — the full record with every table (also written to
``gpurun_out/bench_detail.json``) — and LAST the compact record (< 4 KB: contract keys, ``roofline``, ``cpu_baseline``,
``parity`` and the headline number of every table; ``compact_line``), which is what the driver parses.  No process group
is created in the default N = 1 run.  Besides the contract keys the full record carries
  ``roofline``      the dominant hand-written HBM-bound kernel (SCA deformable-sampling
                    forward), timed live with HIP events on its launch stream;
  ``cpu_baseline``  the oracle's pure-PyTorch CPU port of the same encoder on the host
                    cores (1 warm-up + 3 runs, median; N = 1 only);
  ``parity``        the GPU output of the benched configuration against that oracle run
                    (the run FAILS when it is outside the stated tolerance);
  ``windows``       the K-step window repeated, min / median;
  ``variants``      the same step in the other arithmetic modes (strict-fp32 GEMMs, the bf16
                    configuration) and forward + backward (base and small4), N = 1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time

# c10d / RCCL warnings go to stderr as lines starting with "[": keep the captured stream free of anything a JSON
# scanner could mistake for a record (set BEFORE torch is imported)
os.environ.setdefault("TORCH_CPP_LOG_LEVEL", "ERROR")
os.environ.setdefault("NCCL_DEBUG", "WARN")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ENC_TOL = 5e-4         # encoder-level fp32 tolerance (tests/test_encoder_gpu.py uses the same)
N_RIGS = 8             # distinct camera-matrix sets cycled through by the steps


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="how many times the K-step window is timed")
    ap.add_argument("--workload", default="base")
    ap.add_argument("--gemm", default=None, choices=["split", "bf16", "native"],
                    help="how the Linear layers run (bevformer_amd.ops.set_gemm_mode); default: the "
                         "package default / BEVMSDA_GEMM")
    ap.add_argument("--value-storage", default="fp32", choices=["fp32", "bf16"],
                    help="storage of the projected value tensors (bf16: written by the projection "
                         "kernel, sampled by the 16-byte-lane bf16 kernel; arithmetic stays fp32)")
    ap.add_argument("--queue", type=int, default=0,
                    help="N > 0: a step is N consecutive frames through PerceptionTransformer.get_bev_features "
                         "(ego-motion shift, prev-BEV rotation, can-bus MLP, flatten + embeddings, encoder), each "
                         "frame's BEV being the next frame's history (BASELINE configs[4] style; eager launches)")
    ap.add_argument("--backward", action="store_true",
                    help="time forward + backward of the encoder (autograd path; one captured HIP graph of the whole "
                         "step unless --graph off; BASELINE configs[2] style)")
    ap.add_argument("--train-mode", action="store_true",
                    help="with --backward: the encoder in train() mode (dropout p = 0.1 active, the reference's training step)")
    ap.add_argument("--first-frame", action="store_true", help="no history BEV (prev_bev=None)")
    ap.add_argument("--static-rig", action="store_true",
                    help="same camera matrices every step (the frame plan is then built once)")
    ap.add_argument("--host-plans", action="store_true",
                    help="frame plans from the torch-op builder with its host syncs instead of the HIP kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra configurations (variants)")
    ap.add_argument("--ddp-eager", action="store_true",
                    help="also time the base training step under DistributedDataParallel on a world-1 process group "
                         "(off by default: the default N = 1 run creates no process group)")
    ap.add_argument("--detail-json", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the full record (variants, per-kernel / per-GEMM tables, multi-GPU model) is also written; "
                         "it is printed as an EARLIER stdout line {\"bench_detail\": ...}; the LAST line is the compact record")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a captured HIP graph in the timed region "
                         "(auto = on, falling back to eager launches if the capture fails; the "
                         "per-kernel HIP-event timings come from an eager pass right before)")
    ap.add_argument("--force-tiling", action="store_true",
                    help="run the multi-GPU schedule (row blocks + RCCL all-gather) even on 1 rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, one rank per GPU (the measured configuration).  gloo: a FUNCTIONAL smoke of "
                         "the N > 1 path on a box with fewer GPUs than ranks — ranks share devices (rank % device_count), "
                         "the all-gather is staged through host memory, eager launches; the line says so and is not a "
                         "measurement")
    ap.add_argument("--tile-layout", default="auto", choices=["auto", "rows", "sectors"],
                    help="BEV tiling over GPUs: contiguous blocks of BEV rows, or angular sectors around the ego vehicle "
                         "(a rank then sees 1-3 cameras instead of 3-4 and projects only those)")
    ap.add_argument("--simulate-rank", default=None, metavar="R,G",
                    help="time rank R's schedule of a G-GPU BEV-tiled job on this one GPU (no process group; the "
                         "all-gather is replaced by a local copy: bev_tiling.BevTiling.simulate)")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="no HIP-event brackets around the kernels (they also keep the value projections off "
                         "their side stream): whole-step timing only")
    ap.add_argument("--row-order", default=None, choices=["raster", "image", "polar"],
                    help="order of the ragged SCA rows inside a camera (default: the encoder's)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch of the roofline kernel from an earlier profile run")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of the sampling-kernel / GEMM / frame-plan launches, recorded on the
    stream the kernels are launched on."""

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

    def summary(self, rows):
        """``rows``: the frame's actual ragged row count (device-side plans report their
        algorithmic bytes as (fixed, per row))."""
        agg = {}
        for tag, s, e, b in self.events:
            if tag.startswith("gemm:"):
                continue
            if isinstance(b, tuple):
                b = b[1] + b[2] * rows
            a = agg.setdefault(tag, [0.0, 0, 0])
            a[0] += s.elapsed_time(e) * 1e-3
            a[1] += 1
            a[2] += b
        return {t: dict(avg_us=a[0] / a[1] * 1e6, launches=a[1], alg_bytes=a[2] / a[1],
                        GBs=(a[2] / a[0] / 1e9) if a[2] else None) for t, a in agg.items()}

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


def _pick_cpu_threads(cores, workload, sd, first_frame):
    """Thread count for the CPU leg, probed ON THE BENCHED WORKLOAD: one layer of the frame (``num_layers=1`` of the same
    inputs and weights; the layers are identical work) at a few thread counts up to ``os.cpu_count()``.  The pure-PyTorch
    fallback is made of many small ops next to a few large ``grid_sample`` / GEMM calls and gets slower again with
    every hardware thread of a 2-socket host (measured: 256 threads -> 135 s per base frame vs ~10 s on 8-16), so the
    fastest count is used for the timed frames — and the ``os.cpu_count()`` figure SURVEY.md §8d names is reported
    beside it (``all_cores``).  Returns (threads, {threads: seconds of one layer})."""
    from bevformer_amd import synthetic as S
    from oracle import bevformer_cpu as O
    q, f, kw = S.make_inputs(workload, seed=0, temporal=not first_frame)
    cands = sorted({c for c in (8, 16, 32, 64) if c <= cores} | {cores})
    seen = {}
    with torch.no_grad():
        torch.set_num_threads(min(cands))
        O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, num_layers=1, **kw)      # pages in the inputs, warms the pool
        for c in cands:
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, num_layers=1, **kw)
            seen[c] = time.perf_counter() - t0
            if c >= 64 and seen[c] > 3.0 * min(seen.values()) and c != cores:
                break                       # (clearly past the optimum: only the all-cores figure is still wanted)
    best = min(seen, key=seen.get)
    return best, seen


def cpu_baseline(workload, sd, first_frame, runs=3):
    """The oracle's CPU port of the encoder on the host cores (bounded sample: frames of the
    same workload; fp32, no_grad; thread count picked by a probe on one layer of this workload; BASELINE.md §2 protocol:
    one warm-up run, then ``runs`` timed runs, median).  Returns (json object, oracle output of the
    frame) — the output is what ``parity`` checks the GPU step against."""
    from bevformer_amd import synthetic as S
    from oracle import bevformer_cpu as O
    cores = os.cpu_count() or 1
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    threads, probe = _pick_cpu_threads(cores, workload, sd, first_frame)
    torch.set_num_threads(threads)
    w = S.WORKLOADS[workload]
    times = []
    with torch.no_grad():
        q, f, kw = S.make_inputs(workload, seed=0, temporal=not first_frame)
        for i in range(1 + runs):
            t0 = time.perf_counter()
            out = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)
            if i:
                times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    Q = w["bev_h"] * w["bev_w"]
    all_cores = None
    if cores in probe:
        # SURVEY.md §8d's protocol figure (torch.set_num_threads(os.cpu_count())): one layer timed, x layers
        all_cores = dict(threads=cores, seconds_one_layer=round(probe[cores], 3),
                         value=Q / (probe[cores] * w["layers"]), unit="BEV queries/s",
                         note=f"one layer of the frame timed at os.cpu_count() threads x {w['layers']} identical layers")
    return dict(value=Q / dt, unit="BEV queries/s", cores=threads, kind="port",
                seconds=dt, runs_s=[round(t, 3) for t in times], protocol="1 warm-up + 3 runs, median",
                host_cpus=cores, all_cores=all_cores,
                thread_probe_one_layer_of_this_workload_s={str(k): round(v, 3) for k, v in probe.items()},
                sample=f"1 frame of {workload} ({w['layers']} layers, {Q} queries, "
                       f"{'no ' if first_frame else ''}history BEV) through oracle/bevformer_cpu.py "
                       "(pure-PyTorch CPU fallback path of the reference, fp32, no_grad)"), out


def jittered_rigs(workload, n, device):
    """(n, Nc, 4, 4) fp32 device tensor of camera matrices: entry 0 is the synthetic rig of
    SURVEY §8d, the others the same rig under a small seeded ego-pose change (yaw ~ 0.6 deg,
    translation ~ 0.2 m) — nuScenes rebuilds lidar2img per sample from per-sample
    sensor2lidar transforms (datasets/nuscenes_dataset.py:126-139)."""
    from bevformer_amd import synthetic as S
    import math
    base = np.asarray(S.make_img_metas(workload)[0]["lidar2img"])
    rng = np.random.default_rng(12)
    out = [base]
    for _ in range(n - 1):
        yaw = rng.normal(0, 0.01)
        T = np.eye(4)
        T[:2, :2] = [[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]]
        T[:3, 3] = rng.normal(0, 0.2, 3)
        out.append(base @ T)
    return torch.tensor(np.stack(out), dtype=torch.float32, device=device)


def parity_report(got, want, tol):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    diff = (got - want).abs()
    lim = tol + tol * want.abs()
    cos = torch.nn.functional.cosine_similarity(got.flatten().double(), want.flatten().double(), dim=0).item()
    return dict(max_abs=diff.max().item(), mean_abs=diff.mean().item(), cos=cos, rtol=tol, atol=tol,
                worst_ratio=(diff / lim).max().item(), ok=bool((diff <= lim).all().item()),
                against="oracle/bevformer_cpu.py on the same weights and frame (rig 0)")


class Config:
    """One encoder + inputs + step function of a (workload, arithmetic mode, direction)."""

    def __init__(self, args, dev, workload, gemm, storage, backward, first_frame, world, tiling, train_mode=False):
        import bevformer_amd
        from bevformer_amd import bev_tiling, ops
        from bevformer_amd import synthetic as S
        self.ops, self.S = ops, S
        self.workload, self.gemm, self.storage, self.backward = workload, gemm, storage, backward
        self.args, self.dev, self.world = args, dev, world
        w = self.w = S.WORKLOADS[workload]
        self.Q = w["bev_h"] * w["bev_w"]
        torch.manual_seed(0)
        enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg(workload)).eval()
        self.sd = S.trained_like_({k: v.clone() for k, v in enc.state_dict().items()}, seed=3)
        enc.load_state_dict(self.sd)
        self.enc = enc.to(dev)
        if train_mode:                  # train(): dropout (p = 0.1 in TSA, SCA and the FFN: the reference's training step) active
            self.enc.train()
        if args.row_order:
            self.enc.sca_row_order = args.row_order
        self.enc.device_plans = not args.host_plans
        if getattr(args, "simulate_rank", None) and world == 1 and not tiling:
            r, g = (int(v) for v in args.simulate_rank.split(","))
            bev_tiling.enable_bev_tiling(self.enc, simulate=(r, g), layout=args.tile_layout)
        elif tiling:
            bev_tiling.enable_bev_tiling(self.enc, layout=args.tile_layout)
        self.q, self.f, self.kw = S.make_inputs(workload, seed=0, temporal=not first_frame, device=dev)
        self.metas0 = self.kw["img_metas"]
        self.rigs = jittered_rigs(workload, N_RIGS, dev)
        self.l2i = self.rigs[0].clone()          # the matrices of the CURRENT frame (graph replays read it)
        self.fresh = not args.static_rig and not args.host_plans
        if self.fresh:
            self.kw = dict(self.kw, img_metas=[dict(lidar2img=self.l2i, img_shape=self.metas0[0]["img_shape"])])
        self.frame = 0
        if backward:
            self.g_out = torch.randn(1, self.Q, 256, device=dev,
                                     generator=torch.Generator(device=dev).manual_seed(1))
            self.qg, self.fg = self.q.clone().requires_grad_(True), self.f.clone().requires_grad_(True)

    def modes(self):
        self.ops.set_gemm_mode(self.gemm)
        self.ops.set_value_storage(torch.bfloat16 if self.storage == "bf16" else torch.float32)

    def next_rig(self):
        """New camera matrices for the next frame (device -> device copy on the launch stream)."""
        if self.fresh:
            self.frame += 1
            self.l2i.copy_(self.rigs[self.frame % N_RIGS])

    def set_rig(self, i):
        self.l2i.copy_(self.rigs[i])

    def encoder_step(self):
        if self.backward:       # fwd + bwd w.r.t. parameters, BEV queries and camera features
            self.enc.zero_grad(set_to_none=True)
            self.qg.grad = self.fg.grad = None
            out = self.enc(self.qg, self.fg, self.fg, **self.kw)
            out.backward(self.g_out)
            return out.detach()
        with torch.no_grad():
            return self.enc(self.q, self.f, self.f, **self.kw)

    def rows(self):
        """Ragged SCA rows of the current frame (one device read; after the timed region) — under BEV tiling THIS rank's rows:
        the sampling launches the kernel timer saw cover the rank's tile, and so must their algorithmic bytes."""
        tile = {}
        if getattr(self.enc, "bev_tiling", None) is not None and self.enc.device_plans:
            from bevformer_amd import bev_tiling
            _, _, qr, cell_perm, _, _ = bev_tiling.rank_tile(self.enc, self.w["bev_h"], self.w["bev_w"], self.dev)
            tile = dict(tile=qr, cell_perm=cell_perm)
        with torch.set_grad_enabled(self.backward):
            plan = self.enc.frame_plan(self.w["bev_h"], self.w["bev_w"], 1, self.kw["img_metas"], self.dev,
                                       torch.float32, **tile)
        if plan.dynamic:
            assert plan.dropped_rows() == 0, "frame plan: rows dropped (row capacity too small)"
            return int(plan.nrows_dev.item())
        return int(plan.row_batch.numel())


HOST_ENQUEUE = []       # seconds the host spent ISSUING each timed window (before its closing fence): launch-bound or not


def timed_windows(cfg, step, fence, steps, windows, graph):
    """``windows`` x (``steps`` steps between fences) -> list of seconds."""
    out = []
    for _ in range(windows):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            cfg.next_rig()
            if graph is not None:
                graph.replay()
            else:
                step()
        HOST_ENQUEUE.append(time.perf_counter() - t0)
        fence()
        out.append(time.perf_counter() - t0)
    return out


def capture(step, fence):
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
        out = step()
    graph.replay()
    fence()
    return graph, out


def make_queue_step(cfg, workload, queue, dev, graph=True):
    """One scene of ``queue`` frames through ``get_bev_features`` with a rolling history BEV (frame i's BEV is
    frame i + 1's history; frame 0 opens the scene): BASELINE configs[4]'s history queue on one GPU.  The
    history driver (bevformer_amd.history.BevHistory = detectors/bevformer.py:236-269) turns ABSOLUTE can-bus
    poses into deltas."""
    import copy as _copy
    import bevformer_amd
    from bevformer_amd import synthetic as S
    from bevformer_amd.history import BevHistory, GraphedBevHistory
    torch.manual_seed(4242)                # (every queue variant gets the same transformer-own weights: one oracle run serves them)
    tr = bevformer_amd.build_transformer(S.transformer_cfg(workload)).eval()
    tr.init_weights()
    tr.encoder = cfg.enc                   # the encoder of `cfg` (trained-like weights, tiling if enabled)
    tr = tr.to(dev)
    mlvl, bq, tkw = S.make_transformer_inputs(workload, seed=0, temporal=False, device=dev)
    tkw.pop("prev_bev")
    queue_metas = []
    for i in range(queue):
        m = _copy.deepcopy(tkw["img_metas"])
        m[0]["scene_token"] = "bench-scene"
        m[0]["can_bus"][:3] = np.array([2.0 * (i + 1), 0.5 * (i + 1), 0.0])
        m[0]["can_bus"][-1] = 4.0 * (i + 1)
        queue_metas.append(m)
    tkw_rest = {k: v for k, v in tkw.items() if k != "img_metas"}

    def bev_fn(f, m, p):
        return tr.get_bev_features(f, bq, prev_bev=p, img_metas=m, **tkw_rest)

    # graph mode: prologue + frame plan + encoder replayed from two captured HIP graphs (first frame of the
    # scene / frame with history); the host runs the history state machine and refreshes the device-side pose
    # and camera matrices (bevformer_amd.history.GraphedBevHistory)
    hist = GraphedBevHistory(bev_fn, mlvl) if graph else BevHistory()

    def step():
        hist.reset()
        out_q = None
        with torch.no_grad():
            for i in range(queue):
                out_q = hist.step(bev_fn, mlvl, queue_metas[i])
        return out_q

    step.launch_mode = "hip graph replay per frame (2 graphs)" if graph else "eager"
    step.oracle_inputs = (tr, mlvl, bq, tkw_rest, queue_metas)
    return step


_QUEUE_ORACLE = {}


def queue_oracle(step, workload):
    """The same scene through the oracle on the host: oracle.get_bev_features under the restated ``forward_test``
    state machine (detectors/bevformer.py:236-269) -> the last frame's BEV (bs, Q, C) on the CPU.  Cached per
    (workload, frames, weights): the fp32 and the bf16 queue variants are checked against the one fp32 oracle run."""
    import copy as _copy
    from bevformer_amd import synthetic as S
    from oracle import bevformer_cpu as O
    tr, mlvl, bq, rest, queue_metas = step.oracle_inputs
    sd = {k: v.detach().float().cpu() for k, v in tr.state_dict().items()}
    key = (workload, len(queue_metas), round(sum(float(v.double().abs().sum()) for v in sd.values()), 3))
    if key in _QUEUE_ORACLE:
        return _QUEUE_ORACLE[key]
    own = {k: v for k, v in sd.items() if not k.startswith(("encoder.", "decoder."))}
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    w = S.WORKLOADS[workload]
    feats = [x.detach().float().cpu() for x in mlvl]
    rest = {k: (v.detach().float().cpu() if torch.is_tensor(v) else v) for k, v in rest.items()}
    bev_h, bev_w = rest.pop("bev_h"), rest.pop("bev_w")

    def fn(f, m, p):
        return O.get_bev_features(own, enc, f, bq.detach().float().cpu(), bev_h, bev_w, img_metas=m, prev_bev=p,
                                  pc_range=S.PC_RANGE, rotate_center=(w["bev_w"] // 2, w["bev_h"] // 2), **rest)
    info = {"prev_bev": None, "scene_token": None, "prev_pos": 0, "prev_angle": 0}
    out = None
    with torch.no_grad():
        for m in queue_metas:
            out = O.forward_test_step(info, fn, feats, _copy.deepcopy(m))
    _QUEUE_ORACLE[key] = out
    return out


def run_variant(args, dev, fence, workload, gemm, storage, backward, steps, windows, want=None, tol=None, queue=0,
                train_mode=False):
    """A secondary configuration on the same line: ms per step (graph replay for forward, eager
    for forward + backward and for the history queue), fresh geometry per step as in the main run.
    ``want``: the oracle's output for this configuration's inputs (forward output of the step; for the queue the
    last frame's BEV, computed here) -> a ``parity`` object; forward + backward steps also report ``roofline_bwd``
    (the SCA operator backward: HIP events around its launches in one extra eager step)."""
    cfg = Config(args, dev, workload, gemm, storage, backward, args.first_frame, 1, False, train_mode=train_mode)
    cfg.modes()
    step = make_queue_step(cfg, workload, queue, dev, graph=args.graph != "off") if queue else cfg.encoder_step
    for _ in range(2):
        step()
    fence()
    graph, note, g_out = None, "eager", None
    if not queue and args.graph != "off":
        # forward + backward too (round 4): the autograd path keeps the ragged row count on the device, so the whole
        # step — frame plan, forward, backward, gradient accumulation into fresh buffers — is one captured graph
        try:
            graph, g_out = capture(step, fence)
            note = "hip graph replay" + (" (forward + backward in one graph)" if backward else "")
        except Exception as e:      # noqa: BLE001
            graph, note = None, f"eager (capture failed: {type(e).__name__}: {str(e)[:100]})"
            torch.cuda.synchronize()
    ts = timed_windows(cfg, step, fence, steps, windows, graph)
    per = [t / steps * 1e3 for t in ts]
    res = dict(workload=workload, gemm=gemm, value_storage=storage, direction="fwd+bwd" if backward else "fwd",
               ms_per_step=statistics.median(per), ms_per_step_min=min(per), steps=steps, windows=windows,
               queries_per_s=cfg.Q * max(1, queue) / (statistics.median(per) * 1e-3), launch_mode=note)
    if train_mode:
        res["mode"] = ("train(): dropout p = 0.1 active in TemporalSelfAttention, SpatialCrossAttention and the FFN (masks drawn "
                       "by torch's generator inside the captured step); `parity`: one eager step against the oracle's "
                       "train() mode fed the same scale tensors")
    if queue:
        res["launch_mode"] = step.launch_mode
        res["frames_per_step"] = queue
        res["note"] = ("get_bev_features over a scene of %d frames with a rolling history BEV (can-bus MLP, shift, "
                       "rotation of the history, encoder) per step" % queue)
        got = step().detach().float().cpu()
        want_q = queue_oracle(step, workload)
        rows = (got - want_q).abs().amax(-1).flatten()
        rep = parity_report(got, want_q, tol)
        # nearest-neighbour rotation of the history: a pixel on a rounding tie may take the neighbouring source row
        # on the GPU (tests/test_history_gpu.py counts them); bound the fraction of rows instead of every element
        rep["rows_over_tol_frac"] = float((rows > 2 * tol).float().mean())
        rep["ok"] = bool(rep["rows_over_tol_frac"] < 5e-3 and rep["cos"] > 0.999)
        rep["against"] = "oracle.get_bev_features under the restated forward_test state machine, same scene"
        res["parity"] = rep
    elif want is not None:
        cfg.set_rig(0)
        eager = cfg.encoder_step()
        res["parity"] = parity_report(eager, want, tol)
        if graph is not None:
            graph.replay()
            fence()
            res["parity"]["graph_replay_equals_eager"] = bool(torch.equal(g_out, eager))
            res["parity"]["ok"] = res["parity"]["ok"] and res["parity"]["graph_replay_equals_eager"]
    if train_mode:
        res["parity"] = train_mode_parity(cfg, workload, tol if tol is not None else ENC_TOL)
    if backward:
        from bevformer_amd import ops
        kt = KernelTimer()
        kt.enabled = True
        ops.set_kernel_timer(kt)
        cfg.encoder_step()
        fence()
        ops.set_kernel_timer(None)
        ks = kt.summary(cfg.rows())
        b = ks.get("sca_bwd")
        if b:
            res["roofline_bwd"] = {"kernel": "msda backward, SCA rows (grad_value sort kernel + grad_loc / grad_attn gather kernel)",
                                   "bound": "hbm", "achieved": b["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": b["GBs"] / HBM_PEAK_GBS, "avg_us": b["avg_us"], "alg_bytes": b["alg_bytes"],
                                   "launches_timed": b["launches"], "traffic": None,
                                   "timing": "HIP events on the launch stream around the operator's backward, one eager step"}
        t = ks.get("tsa_bwd")
        if t:
            res["roofline_bwd_tsa"] = {"achieved": t["GBs"], "frac": t["GBs"] / HBM_PEAK_GBS, "avg_us": t["avg_us"],
                                       "alg_bytes": t["alg_bytes"], "unit": "GB/s"}
    del cfg, graph
    torch.cuda.empty_cache()
    return res


def train_mode_parity(cfg, workload, tol):
    """Parity object of a train() mode step (round 6): ONE eager step on rig 0 with ``train_ops.dropout_scale`` — the one
    place the fast path draws its dropout scale tensors (TSA output, SCA output, FFN hidden, FFN output per layer, in that
    order) — recording what it hands out, then the oracle's train() mode on the host with exactly those tensors
    (``O.encoder_forward(dropout_scales=...)``, pinned bit-exact against the reference's own files in train():
    tests/test_oracle_vs_reference.py::test_restatement_in_train_mode_is_bit_exact)."""
    from bevformer_amd import synthetic as S
    from bevformer_amd import train_ops
    from oracle import bevformer_cpu as O
    drawn, real = [], train_ops.dropout_scale

    def recording(shape, p, device):
        t = real(shape, p, device)
        drawn.append(t)
        return t
    train_ops.dropout_scale = recording
    try:
        cfg.set_rig(0)
        got = cfg.encoder_step()
    finally:
        train_ops.dropout_scale = real
    L = cfg.w["layers"]
    if len(drawn) != 4 * L:
        return dict(ok=False, error=f"{len(drawn)} dropout draws in the step, expected {4 * L} (4 sites x {L} layers): "
                                    "the step did not take the chain kernels")
    sd = {k: v.detach().float().cpu() for k, v in cfg.sd.items()}
    q, f, kw = S.make_inputs(workload, seed=0, temporal=not cfg.args.first_frame)
    with torch.no_grad():
        want = O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, dropout_scales=[t.cpu() for t in drawn], **kw)
    rep = parity_report(got, want, tol)
    rep["against"] = ("oracle/bevformer_cpu.py in train() mode with the step's own dropout scale tensors "
                      f"({len(drawn)} draws, {sum(float((t == 0).float().mean()) for t in drawn) / len(drawn):.3f} of the elements dropped)")
    return rep


def run_ddp_eager(args, dev, fence, gemm, steps=3, windows=3):
    """The training step the reference's parallel strategy permits (``MMDistributedDataParallel`` = torch DDP, one process
    per GPU: bevformer/apis/mmdet_train.py:75-79): the base encoder wrapped in ``DistributedDataParallel`` on a world-1 RCCL
    group, forward + backward launched EAGERLY (the reducer's hooks and bucket all-reduces are host-driven: the step cannot
    be one captured HIP graph), beside the same eager step without the wrapper.  The fast path's flattened parameters,
    merged-gradient views and gradient arena run under the reducer here (gradient equality with two ranks:
    tests/test_ddp_gpu.py)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    own = not dist.is_initialized()
    backend = "nccl"
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        except Exception:       # noqa: BLE001
            dist.init_process_group("gloo", rank=0, world_size=1)
            backend = "gloo"
    try:
        cfg = Config(args, dev, "base", gemm, "fp32", True, args.first_frame, 1, False)
        cfg.modes()

        def timed(step):
            for _ in range(2):
                cfg.next_rig()
                step()
            fence()
            ts = timed_windows(cfg, step, fence, steps, windows, None)
            return statistics.median(ts) / steps * 1e3
        plain = timed(cfg.encoder_step)
        ddp = DDP(cfg.enc, device_ids=[dev.index], broadcast_buffers=False)

        def ddp_step():
            ddp.zero_grad(set_to_none=True)
            cfg.qg.grad = cfg.fg.grad = None
            out = ddp(cfg.qg, cfg.fg, cfg.fg, **cfg.kw)
            out.backward(cfg.g_out)
            return out.detach()
        wrapped = timed(ddp_step)
        ok = all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in cfg.enc.parameters())
        return dict(workload="base", gemm=gemm, value_storage="fp32", direction="fwd+bwd",
                    launch_mode="eager (DistributedDataParallel: reducer hooks + bucket all-reduce, world 1, %s)" % backend,
                    ms_per_step=wrapped, ms_per_step_eager_without_ddp=plain, steps=steps, windows=windows,
                    queries_per_s=cfg.Q / (wrapped * 1e-3), all_parameter_gradients_finite=ok,
                    note="what the reference's training wrapper costs around the fast path: the same step as fwd_bwd_base, "
                         "launched eagerly under DDP (no graph); gradient equality across 2 ranks: tests/test_ddp_gpu.py")
    finally:
        if own:
            dist.destroy_process_group()


def oracle_frame(workload, first_frame):
    """Oracle output of the synthetic frame of ``workload`` on the host (what the variants of other workloads are
    checked against)."""
    from bevformer_amd import synthetic as S
    from oracle import bevformer_cpu as O
    import bevformer_amd
    torch.manual_seed(0)
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg(workload)).eval()
    sd = S.trained_like_({k: v.clone() for k, v in enc.state_dict().items()}, seed=3)
    q, f, kw = S.make_inputs(workload, seed=0, temporal=not first_frame)
    with torch.no_grad():
        return O.encoder_forward(sd, q, f, pc_range=S.PC_RANGE, **kw)


def multi_gpu_model(args, dev, fence, gemm, t1_ms, replicated_us, worlds=(2, 4, 8), storage="fp32"):
    """What the BEV-tiled schedule costs per rank, measured on THIS GPU one rank at a time
    (bev_tiling.BevTiling.simulate: every kernel of rank r's step, the all-gather replaced by a local copy),
    plus a stated model of the all-gather.  Strong-scaling efficiency modelled from it = t1 / (G * T_G) with
    T_G = max over ranks + all-gather.  NOT a multi-GPU measurement: one GPU per box here."""
    from bevformer_amd import bev_tiling
    LINK_GBS, LAT_US = 50.0, 20.0
    cfg = Config(args, dev, "base", gemm, storage, False, args.first_frame, 1, False)
    cfg.modes()
    out = {"arithmetic": {"gemm": gemm, "value_storage": storage},
           "assumptions": {"xgmi_link_GBs_per_direction": LINK_GBS, "collective_latency_us": LAT_US,
                           "all_gather": "direct: every rank sends its (Q/G, 256) fp32 shard to the G - 1 peers over "
                                         "min(G - 1, 7) links in parallel"},
           "t1_ms": t1_ms, "replicated_value_projections_us": replicated_us,
           "status": "per-rank times measured on one GPU; collective modelled; unmeasured on multi-GPU hardware"}
    def one_layout(layout):
        res = {}
        for G in worlds:
            per_rank, cams = [], []
            for r in range(G):
                bev_tiling.enable_bev_tiling(cfg.enc, simulate=(r, G), layout=layout)
                for _ in range(2):
                    cfg.encoder_step()
                fence()
                graph = None
                if args.graph != "off":
                    try:
                        graph, _ = capture(cfg.encoder_step, fence)
                    except Exception:      # noqa: BLE001
                        graph = None
                        torch.cuda.synchronize()
                ts = timed_windows(cfg, cfg.encoder_step, fence, 10, 3, graph)
                per_rank.append(statistics.median(ts) / 10 * 1e3)
                seg = getattr(cfg.enc, "_last_segments", None)
                cams.append(int(((seg[0][1:] - seg[0][:-1]) > 0).sum()) if seg is not None else None)
                del graph
            shard = cfg.Q / G * 256 * 4
            ag_us = LAT_US + shard * (G - 1) / (min(G - 1, 7) * LINK_GBS * 1e9) * 1e6
            T = max(per_rank) + ag_us * 1e-3
            res[str(G)] = dict(per_rank_ms=[round(p, 4) for p in per_rank], cameras_projected_per_rank=cams,
                               all_gather_model_us=ag_us, step_ms=T,
                               queries_per_s=cfg.Q / (T * 1e-3), efficiency=t1_ms / (G * T),
                               amdahl_bound_efficiency=t1_ms / (G * (replicated_us * 1e-3 + (t1_ms - replicated_us * 1e-3) / G))
                               if replicated_us else None)
        return res

    # the layout that would run (auto: rows at 2 ranks, sectors from 3 on) and, beside it, both fixed layouts
    out["layout"] = args.tile_layout
    fixed = {lay: one_layout(lay) for lay in ("rows", "sectors")}
    for G in worlds:
        lay = args.tile_layout if args.tile_layout != "auto" else ("sectors" if G >= 3 else "rows")
        out[str(G)] = dict(fixed[lay][str(G)], layout=lay)
    out["by_layout"] = fixed
    bev_tiling.disable_bev_tiling(cfg.enc)
    del cfg
    torch.cuda.empty_cache()
    return out


def strong_scaling_note(line, replicas, world):
    """What the N > 1 line says about its own strong scaling: the single-frame time of one GPU (measured here as
    ``frames_in_parallel``: every rank running whole untiled frames), the efficiency t1 / (N * T_N) that follows, and the
    Amdahl bound of this schedule — the two value projections (camera features, history BEV) are REPLICATED on every
    rank because every rank samples all of their output, and sharding + all-gathering them costs more than recomputing
    (~0.5 GB of bf16 values over ~350 GB/s of xGMI ~ 1.4 ms against 0.5 ms of projection)."""
    out = {"north_star_target": 0.85,
           "statement": "0.85 strong-scaling efficiency at 8 GPUs is NOT reachable with exact semantics on a ~4 ms frame: "
                        "the replicated value projections alone bound it (amdahl_bound_efficiency), and the per-rank "
                        "remainder is latency-bound at tile size; the throughput mode of this path across GPUs is "
                        "frames_in_parallel (one frame stream per GPU, no exchange)"}
    gs = line.get("gemms") or {}
    rep = sum(gs.get("per_tag", {}).get(t, {}).get("avg_us", 0.0) for t in ("sca_value_proj", "tsa_value_proj")) * 1e-3
    if replicas and "ms_per_step" in replicas:
        t1 = replicas["ms_per_step"]
        out["t1_ms_one_untiled_frame_per_gpu"] = t1
        out["efficiency_t1_over_N_TN"] = t1 / (world * line["ms_per_step"])
        if rep:
            out["replicated_value_projections_ms_per_rank"] = rep
            out["amdahl_bound_efficiency"] = t1 / (world * (rep + (t1 - rep) / world)) if t1 > rep else None
    return out


def l1_path(w, rows, avg_us, storage):
    """Gather volume of one SCA sampling launch against the L1 data path (64 B/clk/CU x 256 CUs x 2.4 GHz)."""
    if not avg_us:
        return None
    heads, points, D = 8, 8, 32
    gather = rows * heads * len(w["shapes"]) * points * 4 * D * (2 if storage == "bf16" else 4)
    peak = 64 * 256 * 2.4e9
    return dict(gather_bytes=gather, achieved_TBs=gather / (avg_us * 1e-6) / 1e12, peak_TBs=peak / 1e12,
                frac=gather / (avg_us * 1e-6) / peak)

COMPACT_LIMIT = 4000        # bytes: the driver keeps an 8,191-byte tail of the stream; the record must sit well inside it


def _r(x, sig=6):
    """Floats to ``sig`` significant digits (the compact record), containers recursively."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_line(line):
    """The driver's record: the contract keys + ``roofline`` + ``cpu_baseline`` + ``parity`` + a few scalars, every
    table (variants, kernels, gemms, multi-GPU model) reduced to its headline numbers.  Always < COMPACT_LIMIT bytes:
    optional keys are dropped from the end of ``optional`` until it fits."""
    c = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline", "dtype", "data")}
    cfg = line.get("config") or {}
    c["config"] = {k: cfg.get(k) for k in ("workload", "global_batch", "parallelism", "value_storage", "sca_row_order",
                                           "sca_rows_per_frame")}
    c["config"]["geometry"] = "frame plan rebuilt every step" if str(cfg.get("geometry", "")).startswith("new") else "static rig"
    r = line.get("roofline") or {}
    c["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us",
                                           "alg_bytes", "launches_timed")}
    if "cpu_baseline" in line:
        cb = line["cpu_baseline"]
        c["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "seconds", "host_cpus")}
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        c["vs_cpu_baseline"] = line.get("vs_cpu_baseline")
    if "parity" in line:
        p = line["parity"]
        c["parity"] = {k: p.get(k) for k in ("ok", "max_abs", "worst_ratio", "rtol", "atol", "graph_replay_equals_eager")}
    c["launch_mode"] = line.get("launch_mode")
    c["host_issue_ms_per_step"] = line.get("host_issue_ms_per_step")
    c["ranks"] = line.get("ranks")
    w = line.get("windows") or {}
    c["windows"] = {k: w.get(k) for k in ("min", "median", "n")}
    optional = []
    if line.get("variants"):
        v = {}
        for name, x in line["variants"].items():
            e = {"ms": x.get("ms_per_step")}
            if "parity" in x:
                e["ok"] = x["parity"].get("ok")
            if "roofline_bwd" in x:
                e["bwd_frac"] = x["roofline_bwd"].get("frac")
            if "error" in x:
                e["error"] = str(x["error"])[:60]
            v[name] = e
        optional.append(("variants", v))
    for key in ("frames_in_parallel", "collective"):
        if line.get(key):
            x = line[key]
            optional.append((key, {k: x[k] for k in ("ms_per_step", "value", "scaling", "us_per_call_max_over_ranks",
                                                     "shard_bytes", "backend", "error") if k in x}))
    if line.get("strong_scaling"):
        x = line["strong_scaling"]
        optional.append(("strong_scaling", {k: x[k] for k in ("north_star_target", "t1_ms_one_untiled_frame_per_gpu",
                                                              "efficiency_t1_over_N_TN", "amdahl_bound_efficiency") if k in x}))
    if line.get("rank_skew"):
        optional.append(("rank_skew_ms", line["rank_skew"].get("max_minus_min_ms")))
    for key in ("multi_gpu_model", "multi_gpu_model_bf16"):
        m = line.get(key)
        if m:
            optional.append((key, {"status": "per-rank times on ONE GPU + modelled all-gather; not a multi-GPU measurement",
                                   **{g: {"layout": m[g].get("layout"), "per_rank_max_ms": max(m[g]["per_rank_ms"]),
                                          "step_ms": m[g]["step_ms"], "efficiency": m[g]["efficiency"]}
                                      for g in ("2", "4", "8") if g in m}}))
    g = line.get("gemms")
    if g:
        optional.append(("gemms", {"total_us_per_step": g.get("total_us_per_step"), "TFLOPs": g.get("TFLOPs"),
                                   "per_tag_avg_us": {t: x["avg_us"] for t, x in g.get("per_tag", {}).items()}}))
    k = line.get("kernels")
    if k:
        optional.append(("kernels_avg_us", {t: x["avg_us"] for t, x in k.items()}))
    if line.get("smoke"):
        optional.insert(0, ("smoke", line["smoke"][:120]))
    c["detail"] = "full record: the stdout line {\"bench_detail\": ...} printed before this one"
    for key, val in optional:
        c[key] = val
    c = _r(c, 5)
    drop = [key for key, _ in optional]
    while len(json.dumps(c, separators=(",", ":"))) > COMPACT_LIMIT and drop:
        c.pop(drop.pop(), None)
    return c


def emit(line, detail_path=None):
    """Full record first (one stdout line, also a file when the directory is writable), the compact record LAST."""
    detail = json.dumps({"bench_detail": line})
    print(detail, flush=True)
    if detail_path:
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as fh:
                fh.write(json.dumps(line, indent=1))
        except OSError:
            pass
    out = json.dumps(compact_line(line), separators=(",", ":"))
    assert len(out) <= COMPACT_LIMIT and "\n" not in out
    print(out, flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU product path)"
    smoke = args.dist_backend == "gloo" and world > 1
    if smoke:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if smoke:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    elif args.force_tiling:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    tiling = world > 1 or args.force_tiling

    import bevformer_amd
    from bevformer_amd import ops
    from bevformer_amd import synthetic as S

    gemm = args.gemm or ops.gemm_mode()
    cfg = Config(args, dev, args.workload, gemm, args.value_storage, args.backward, args.first_frame, world, tiling,
                 train_mode=bool(args.backward and args.train_mode))
    cfg.modes()
    timer = KernelTimer()
    if not args.no_kernel_timers:
        ops.set_kernel_timer(timer)
        ops.set_gemm_timer(timer.gemm)
    w, Q = cfg.w, cfg.Q

    queue_step = make_queue_step(cfg, args.workload, args.queue, dev, graph=args.graph != "off") if args.queue > 0 else None

    def step():
        if queue_step is not None:
            return queue_step()
        return cfg.encoder_step()

    def fence():
        torch.cuda.synchronize()
        if tiling:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        cfg.next_rig()
        step()
    fence()

    # Timed region: replays of ONE captured HIP graph of the whole step (frame plan + ~100 launches;
    # with N > 1 the collective is captured too), per-kernel durations from HIP events
    # around every sampling / GEMM launch of two eager steps right before; or (--graph off, or
    # a failed capture) eager launches with the events recorded inside the timed region.
    graph = None
    # (gloo smoke: the host-staged all-gather cannot be captured)
    use_graph = args.graph in ("on", "auto") and args.queue == 0 and not smoke
    graph_note = "eager" if not smoke else "eager (gloo smoke: the all-gather is staged through host memory)"
    if use_graph:
        timer.enabled = True            # kernel durations from an eager pass (events cannot
        for _ in range(2):              # bracket nodes inside a graph replay)
            cfg.next_rig()
            step()
        fence()
        timer.enabled = False
        try:
            graph, g_out = capture(step, fence)
            graph_note = "hip graph replay"
        except Exception as e:          # noqa: BLE001 — any capture problem: measure eagerly
            graph = None
            graph_note = f"eager (graph capture failed: {type(e).__name__}: {str(e)[:120]})"
            torch.cuda.synchronize()
    if graph is None and queue_step is not None and getattr(queue_step, "launch_mode", "eager") != "eager":
        # the queue replays its own graphs: kernel durations from one eager pass of the same frames
        eager_q = make_queue_step(cfg, args.workload, args.queue, dev, graph=False)
        timer.enabled = True
        eager_q()
        fence()
        timer.enabled = False
        graph_note = queue_step.launch_mode
    elif graph is None:
        timer.enabled = not timer.events
    del HOST_ENQUEUE[:]
    ts = timed_windows(cfg, step, fence, args.steps, 1, graph)       # window 0 carries the in-region events
    timer.enabled = False
    ts += timed_windows(cfg, step, fence, args.steps, max(0, args.windows - 1), graph)
    rank_ms = None
    if world > 1:
        t = torch.tensor(ts, device=dev, dtype=torch.float64)
        try:            # every rank's own clock around the same windows -> per-rank step time and skew on the line
            allt = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            rank_ms = [statistics.median(x.tolist()) / args.steps * 1e3 for x in allt]
        except Exception:       # noqa: BLE001
            rank_ms = None
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts = [float(v) for v in t.tolist()]
    dt = statistics.median(ts)
    out = g_out if graph is not None else step()
    assert torch.isfinite(out).all()

    # geometry alone: the frame plan (camera-matrix copy + the two plan kernels), HIP events
    geometry_ms = None
    if cfg.fresh and args.queue == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.no_grad():
            fence()
            ev[0].record()
            for _ in range(10):
                cfg.next_rig()
                cfg.enc.frame_plan(w["bev_h"], w["bev_w"], 1, cfg.kw["img_metas"], dev, torch.float32)
            ev[1].record()
            fence()
        geometry_ms = ev[0].elapsed_time(ev[1]) / 10

    if rank == 0:
        rows_per_frame = cfg.rows()
        ks = timer.summary(rows_per_frame)
        gs = timer.gemm_summary()
        if gs is not None:
            steps_timed = max(1, ks.get("sca_fwd", {"launches": w["layers"]})["launches"] // w["layers"])
            gs["total_us_per_step"] = gs.pop("seconds") / steps_timed * 1e6
            # matrix-core utilisation: the split kernel issues 3 bf16 MFMA products per algorithmic product
            if ops.gemm_mode() != "native":
                gs["frac_of_bf16_mfma_peak_2500"] = gs["TFLOPs"] * (3 if ops.gemm_mode() == "split" else 1) / 2500.0
        dom = ks.get("sca_fwd") or next(iter(ks.values()), None)
        if dom is None:         # --no-kernel-timers: whole-step timing only
            dom = dict(GBs=float("nan"), avg_us=None, alg_bytes=None, launches=0)
        traffic = None
        # (the profile was taken on the untiled frame: a tiled rank's launches cover its tile only — no counter figure for them)
        if os.path.exists(args.traffic_json) and not tiling:
            try:
                tj = json.load(open(args.traffic_json))
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from profile_traffic import kernel_sources_sha
                sha = kernel_sources_sha()
                fresh = tj.get("_kernel_sources_sha") == sha
                traffic = dict(bytes_per_launch=tj.get(args.workload, {}).get("sca_fwd") if fresh else None,
                               source=tj.get("_source"), kernel_sources_sha=sha, profile_kernel_sources_sha=tj.get("_kernel_sources_sha"),
                               note="rocprofv3 PMC passes of an EARLIER run of this bench (not this run)" if fresh else
                                    "STALE: the sampling kernels' sources changed since the profile was collected "
                                    "(python tools/profile_traffic.py --config base_fwd on the GPU box, then --install)")
            except Exception:       # noqa: BLE001
                traffic = None
        with torch.set_grad_enabled(args.backward):
            row_order_used = cfg.enc.row_order()
        per = [t / args.steps * 1e3 for t in ts]
        gemm_desc = {"split": "hand-written MFMA kernel, fp32 operands split into 2 bf16 terms, "
                              "3 bf16 MFMA products per fp32 product, fp32 accumulate",
                     "bf16": "hand-written MFMA kernel, operands rounded to bf16, fp32 accumulate",
                     "native": "hipBLASLt fp32 (torch.nn.functional.linear)"}[ops.gemm_mode()]
        line = {
            "metric": "BEV-encoder queries/sec (200x200 BEV, 6 cams, 4 lvls)" if args.workload == "base"
            else f"BEV-encoder queries/sec ({args.workload})",
            "value": Q * max(1, args.queue) * args.steps / dt, "unit": "BEV queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            # the host's share: time to ISSUE a step (median over windows; the fence is not in it).  Well under ms_per_step
            # = the host runs ahead of the GPU and the step is device-bound
            "host_issue_ms_per_step": statistics.median(HOST_ENQUEUE[:len(ts)]) / args.steps * 1e3 if HOST_ENQUEUE else None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # arithmetic of the path: fp32 storage / sampling / softmax / LayerNorm / accumulation always;
            # GEMM products from bf16x3-split fp32 operands ("f32/bf16x3"), bf16-rounded operands ("bf16")
            # or hipBLASLt fp32 ("f32")
            "dtype": {"split": "f32/bf16x3", "bf16": "bf16", "native": "f32"}[ops.gemm_mode()],
            "data": "synthetic",
            "config": {"workload": f"bevformer_{args.workload} BEV encoder {'forward + backward' if args.backward else 'forward'}, "
                                   f"{args.queue if args.queue else 1} frame{'s' if args.queue > 1 else ''}/step"
                                   f"{' through get_bev_features with a rolling history BEV' if args.queue else ''}, "
                                   f"{w['bev_h']}x{w['bev_w']} queries, 6 cams, {len(w['shapes'])} levels, "
                                   f"{w['layers']} layers, " + ("frame 0 without history, frames 1.. with the previous frame's BEV"
                                                           if args.queue else
                                                           ('first frame (no history)' if args.first_frame else 'with history BEV')),
                       "geometry": ("new camera matrices every step: frame plan (projection, visibility, ragged rows) "
                                    "rebuilt by the HIP plan kernels inside the timed step" if cfg.fresh else
                                    "same camera matrices every step: frame plan built once (static rig)"),
                       "sca_row_order": row_order_used,
                       "value_storage": args.value_storage,
                       "gemm": gemm_desc,
                       "global_batch": 1, "parallelism": (f"bev-{cfg.enc.bev_tiling.layout if getattr(cfg.enc, 'bev_tiling', None) is not None else 'row'}-tiles x{world}"
                                       if world > 1 else "single GPU"),
                       # (BEV tiling: the rows of rank 0's tile — what its sampling launches, and the roofline below, cover)
                       "sca_rows_per_frame": rows_per_frame},
            "windows": {"ms_per_step": [round(p, 4) for p in per], "min": min(per), "median": statistics.median(per),
                        "n": len(per), "note": "ms_per_step / value = the median window"},
            "geometry_ms": geometry_ms,
            "roofline": {"kernel": "msda_fwd (SCA sampling, ragged rows)", "bound": "hbm",
                         "achieved": dom["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["GBs"] / HBM_PEAK_GBS,
                         # HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950
                         # correction): collected by `tools/profile_traffic.py` in separate rocprofv3 --pmc passes of
                         # THIS bench command on an earlier visit (profiles/traffic.json) — a bench run cannot collect
                         # counters on itself; null when no profile of this workload is on disk
                         "traffic": (traffic or {}).get("bytes_per_launch"),
                         "traffic_provenance": (None if not (traffic or {}).get("bytes_per_launch") else
                                                f"{os.path.relpath(args.traffic_json, ROOT)} <- {(traffic or {}).get('source')}: "
                                                "rocprofv3 --pmc passes of an earlier run of this same bench command, "
                                                "not collected in this run"),
                         "traffic_from_profile": traffic,
                         "avg_us": dom["avg_us"], "alg_bytes": dom["alg_bytes"],
                         "launches_timed": dom["launches"],
                         "timing": "HIP events on the launch stream, " +
                                   ("every launch of the first timed window" if graph is None
                                    else "eager pass right before the timed region"),
                         # what actually bounds the kernel (counters in profiles/r2): the lanes
                         # request rows x heads x levels x points x 4 taps x (32 channels x bytes per channel)
                         # through the CUs' vector-memory path, 64 B/clk/CU
                         "l1_path": l1_path(w, rows_per_frame, dom["avg_us"], args.value_storage)},
            "launch_mode": graph_note,
            "kernels": ks,
            "gemms": gs,
        }
        ok = True
        if world == 1 and not args.no_cpu_baseline:
            cb, want = cpu_baseline(args.workload, cfg.sd, args.first_frame)
            line["cpu_baseline"] = cb
            line["vs_cpu_baseline"] = line["value"] / cb["value"]
            if not args.backward and args.queue == 0:
                # what was timed is also what is checked: the benched configuration on rig 0 against
                # the oracle output of the same frame
                cfg.set_rig(0)
                tol = ENC_TOL if (args.value_storage == "fp32" and ops.gemm_mode() != "bf16") else 5e-2
                eager = step()
                line["parity"] = parity_report(eager, want, tol)
                if graph is not None:       # what was TIMED (the replayed graph) against what is checked (the eager step)
                    graph.replay()
                    fence()
                    line["parity"]["graph_replay_equals_eager"] = bool(torch.equal(g_out, eager))
                    line["parity"]["ok"] = line["parity"]["ok"] and line["parity"]["graph_replay_equals_eager"]
                ok = line["parity"]["ok"]
            if not args.no_variants and not args.backward and args.queue == 0 and not tiling \
                    and args.workload == "base":
                ops.set_kernel_timer(None)
                ops.set_gemm_timer(None)
                v = {}
                v["native_fp32"] = run_variant(args, dev, fence, "base", "native", "fp32", False, 10, 3, want, ENC_TOL)
                v["bf16"] = run_variant(args, dev, fence, "base", "bf16", "bf16", False, 10, 3, want, 5e-2)
                want4 = oracle_frame("small4", args.first_frame)
                v["fwd_bwd_base"] = run_variant(args, dev, fence, "base", gemm, "fp32", True, 3, 3, want, ENC_TOL)
                # the same step in train() mode: parity against the oracle's train() mode fed the step's own dropout scale tensors
                v["fwd_bwd_base_train_mode"] = run_variant(args, dev, fence, "base", gemm, "fp32", True, 3, 3, tol=ENC_TOL,
                                                           train_mode=True)
                v["fwd_bwd_small4"] = run_variant(args, dev, fence, "small4", gemm, "fp32", True, 5, 3, want4, ENC_TOL)
                v["fwd_bwd_small4_bf16"] = run_variant(args, dev, fence, "small4", "bf16", "bf16", True, 5, 3, want4, 5e-2)
                # the reference-true bevformer_small shape set (ONE level (23, 40), 3 layers, 150 x 150 queries:
                # projects/configs/bevformer/bevformer_small.py:41-43,88) and BASELINE configs[1] (bevformer_tiny forward)
                want_s = oracle_frame("small", args.first_frame)
                v["fwd_bwd_small"] = run_variant(args, dev, fence, "small", gemm, "fp32", True, 5, 3, want_s, ENC_TOL)
                v["fwd_bwd_small_bf16"] = run_variant(args, dev, fence, "small", "bf16", "bf16", True, 5, 3, want_s, 5e-2)
                v["fwd_small"] = run_variant(args, dev, fence, "small", gemm, "fp32", False, 10, 3, want_s, ENC_TOL)
                v["fwd_tiny"] = run_variant(args, dev, fence, "tiny", gemm, "fp32", False, 20, 3,
                                            oracle_frame("tiny", args.first_frame), ENC_TOL)
                if args.ddp_eager:
                    try:
                        v["fwd_bwd_base_ddp_eager"] = run_ddp_eager(args, dev, fence, gemm)
                    except Exception as e:      # noqa: BLE001 — never lose the line to the extra variant
                        v["fwd_bwd_base_ddp_eager"] = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
                v["queue4_bf16"] = run_variant(args, dev, fence, "base", "bf16", "bf16", False, 3, 3, tol=5e-2, queue=4)
                # the same 4-frame queue in the headline arithmetic (fp32 storage, split-bf16 GEMMs) against the same
                # oracle run, at twice the single-frame tolerance (four chained frames)
                v["queue4_fp32"] = run_variant(args, dev, fence, "base", gemm, "fp32", False, 3, 3, tol=2 * ENC_TOL, queue=4)
                line["variants"] = v
                rep = None
                if gs is not None:
                    rep = sum(gs["per_tag"][t]["avg_us"] for t in ("sca_value_proj", "tsa_value_proj") if t in gs["per_tag"])
                line["multi_gpu_model"] = multi_gpu_model(args, dev, fence, gemm, line["ms_per_step"], rep)
                # the same model in the arithmetic BASELINE configs[4] names (bf16 value storage, bf16-input GEMMs): the
                # replicated projections are MFMA work, a third of it there, so the schedule's Amdahl bound moves
                line["multi_gpu_model_bf16"] = multi_gpu_model(args, dev, fence, "bf16", v["bf16"]["ms_per_step"], None,
                                                               storage="bf16")
                line["native_fp32_ms_per_step"] = v["native_fp32"]["ms_per_step"]
                ok = ok and all(x["parity"]["ok"] for x in v.values() if "parity" in x)
    else:
        line, ok = None, True
    # N > 1: the same GPUs as independent frame streams — every rank runs whole, untiled frames, no exchange (the
    # reference's own data parallelism, bevformer/apis/mmdet_train.py:75-79) — measured after the tiled schedule,
    # all ranks take part.  Reported beside the strong-scaling `value`, never instead of it.
    replicas = None
    if tiling and not args.backward and args.queue == 0:
        try:
            from bevformer_amd import bev_tiling as _bt
            _bt.disable_bev_tiling(cfg.enc)
            ops.set_kernel_timer(None)
            ops.set_gemm_timer(None)
            for _ in range(2):
                cfg.next_rig()
                cfg.encoder_step()
            fence()
            g2 = None
            if args.graph != "off":
                try:
                    g2, _ = capture(cfg.encoder_step, fence)
                except Exception:       # noqa: BLE001
                    g2 = None
                    torch.cuda.synchronize()
            ts2 = timed_windows(cfg, cfg.encoder_step, fence, args.steps, min(3, args.windows), g2)
            if world > 1:
                t2 = torch.tensor(ts2, device=dev, dtype=torch.float64)
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                ts2 = [float(v) for v in t2.tolist()]
            dt2 = statistics.median(ts2)
            replicas = dict(scaling="weak", frames_per_step=world, ms_per_step=dt2 / args.steps * 1e3,
                            value=Q * world * args.steps / dt2, unit="BEV queries/s",
                            note="one untiled frame per GPU per step, no collective")
        except Exception as e:          # noqa: BLE001 — never lose the main line to the extra one
            replicas = dict(error=f"{type(e).__name__}: {str(e)[:160]}")
    # the exchange step alone (N > 1): one all-gather of the (Q / N, 256) fp32 shards, HIP events around 20 calls
    collective = None
    if world > 1:
        try:
            from bevformer_amd import bev_tiling as _bt2
            blocks = _bt2.row_blocks(w["bev_h"], world)
            h0, h1 = blocks[rank]
            shard = torch.randn(1, (h1 - h0) * w["bev_w"], 256, device=dev)
            for _ in range(3):
                _bt2.all_gather_rows(shard, blocks, w["bev_w"])
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _bt2.all_gather_rows(shard, blocks, w["bev_w"])
            e1.record()
            fence()
            t_ag = torch.tensor([e0.elapsed_time(e1) / 20 * 1e3], device=dev, dtype=torch.float64)
            dist.all_reduce(t_ag, op=dist.ReduceOp.MAX)
            collective = dict(backend="nccl (RCCL)" if not smoke else "gloo, staged through host memory (smoke)",
                              ranks=dist.get_world_size(), op="all_gather_into_tensor",
                              shard_bytes=int(shard.numel() * 4), us_per_call_max_over_ranks=float(t_ag.item()),
                              calls_per_step=1 if not args.first_frame else w["layers"])
        except Exception as e:          # noqa: BLE001
            collective = dict(error=f"{type(e).__name__}: {str(e)[:160]}")
    if line is not None:
        line["ranks"] = world
        if rank_ms is not None:
            line["rank_skew"] = dict(per_rank_ms_per_step=rank_ms, max_minus_min_ms=max(rank_ms) - min(rank_ms),
                                     note="each rank's median over the same barrier-bracketed windows; ms_per_step is the MAX")
        if collective is not None:
            line["collective"] = collective
    if line is not None and replicas is not None:
        line["frames_in_parallel"] = replicas
    if line is not None and world > 1:
        line["strong_scaling"] = strong_scaling_note(line, replicas, world)
    if line is not None and smoke:
        line["smoke"] = (f"NOT A MEASUREMENT: {world} ranks on {torch.cuda.device_count()} GPU(s) over gloo with the "
                         "all-gather staged through host memory — a functional run of the N > 1 bench path (ranks, "
                         "collective, frames_in_parallel, JSON line) on a box without enough devices for RCCL")
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
        emit(line, args.detail_json)
    if not ok:
        raise SystemExit("bench: parity check against the oracle FAILED (see the `parity` objects)")


if __name__ == "__main__":
    main()
