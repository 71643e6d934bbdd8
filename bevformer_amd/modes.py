"""Execution modes of the host layer (``ops``): which kernel family, arithmetic and storage a call uses.

One ``Modes`` object holds every switch.  ``current()`` is what a call sees: the innermost ``using(...)`` block of
the CALLING THREAD, else the process-wide defaults (initialised from ``BEVMSDA_*`` environment variables once, at
import; the ``ops.set_*`` functions edit those defaults).  The autograd Functions of ``ops`` snapshot ``current()``
in ``forward`` and re-activate it in ``backward`` — the autograd engine runs backward on its own thread, which
would otherwise see the process defaults instead of the caller's modes.  The C library keeps no state: every
switch travels in a descriptor field of the call (include/bevmsda.h)."""
import contextlib
import copy
import os
import threading

import torch

GEMM_MODES = ("split", "bf16", "native")
GEMM_KERNELS = (None, "first", "first64", "pipe", "panel", "panel64", "panel128", "panel64w2", "panel64w6") + tuple(
    # A/B knobs of the row-panel epilogue (tools/gemm_epilogue_ab.py): e1 dripping stores, e2 weight fragments 4 steps
    # ahead, e3 both, e4 the round-4 epilogue (bias loaded per piece)
    f"panel{bm}e{v}" for bm in (64, 128) for v in (1, 2, 3, 4)) + tuple(
    # ... and of the phase skew of its column sweep (n x 1024 clocks)
    f"panel{bm}s{n}" for bm in (64, 128) for n in (0, 1, 2, 3, 4, 6, 8, 12)) + ("panel128d2", "panel128d4")    # d2 / d4: one wavefront per SIMD, dripping stores


class Modes:
    __slots__ = ("value_storage", "fused", "fused_train", "gemm", "gemm_variant", "gemm_pack", "train_forward_mfma",
                 "gemm_kernel", "ln_fuse", "wgrad", "bf16_lanes8", "fused_wpe", "fused_lds_pad_kb", "chain_shape", "grad_thread", "train_chain", "wgrad_workgroups", "wgrad_variant", "stack_free", "weight_views", "flatten_params", "fused_save", "chain_backward", "grad_arena", "overlap_value_proj", "use_grad_arena", "fused_spec", "fused_capacity_launch", "graph_repack", "tsa_seam", "chain_gather_all", "plan_on_side")

    def __init__(self):
        env = os.environ.get
        self.value_storage = torch.float32      # storage of the projected value tensors sampled by the kernels
        self.fused = True                        # fused softmax + location + sampling kernel on the no-grad path
        self.fused_train = env("BEVMSDA_FUSED_TRAIN", "1") == "1"   # ... and under autograd (three-step backward)
        self.gemm = env("BEVMSDA_GEMM", "split")                    # split | bf16 | native
        self.gemm_variant = int(env("BEVMSDA_GEMM_VARIANT")) if env("BEVMSDA_GEMM_VARIANT") else None  # None, 0, 12
        self.gemm_pack = env("BEVMSDA_GEMM_PACK", "1") == "1"       # pre-split weight images
        self.train_forward_mfma = env("BEVMSDA_TRAIN_FWD_MFMA", "1") == "1"
        k = env("BEVMSDA_GEMM_KERNEL", "")
        self.gemm_kernel = k if k in GEMM_KERNELS[1:] else None     # None = by measurement (ops._panel_covers)
        self.ln_fuse = env("BEVMSDA_FUSE_LN", "1") == "1"           # residual + LayerNorm in the projection's epilogue
        self.wgrad = env("BEVMSDA_WGRAD", "1") == "1"               # weight gradients on the TN MFMA kernel
        self.bf16_lanes8 = env("BEVMSDA_BF16_LANES8") is not None   # benchmark knob: 8-byte-lane bf16 kernels
        self.fused_wpe = int(env("BEVMSDA_FUSED_WPE", "0"))         # benchmark knob: register budget of the fused kernel
        # fused sampling kernels with compile-time head / level counts (msda_d32.h LC / MC), the library's reserved[5]: 0 =
        # default (TemporalSelfAttention's shape specialised at 128 registers), 1 = generic kernels only, 2 = TSA's at 64
        # registers, 3 = SpatialCrossAttention's shape specialised too (A/B knobs; profiles/r5), 4 = TSA's shape on the
        # resident, software-pipelined grid (round 6: 3 % faster, twice the L2 misses — opt-in; profiles/r6x), 5 = TSA's shape with
        # each 16 x 8 tile's tap lines staged in LDS (round 6: bit-equal, level with the default — opt-in; DESIGN K1-LDS)
        self.fused_spec = int(env("BEVMSDA_FUSED_SPEC", "0"))
        # sampling launches over a device-side row count: True = ONE launch sized by the row CAPACITY (surplus workgroups return
        # on their first instruction), False = a launch sized by the host's hint + a small strided tail launch for rows beyond
        # it, "auto" (default) = the capacity launch when it has at most FUSED_CAPACITY_AUTO_ROWS surplus rows (tiles of the BEV
        # grid, small grids: the surplus workgroups cost less than the tail launch's 5 us; at the base grid's 190,000 surplus
        # rows the two are level — profiles/r6z/r6zz_capacity_launch_ab.txt)
        self.fused_capacity_launch = {"0": False, "1": True}.get(env("BEVMSDA_FUSED_CAPACITY", "auto"), "auto")
        # A/B knob: re-pack the weight images of trainable parameters inside EVERY captured graph (round 4's behaviour; the
        # default re-packs only in graphs captured with grad mode on: ops.images._cache_ok)
        self.graph_repack = env("BEVMSDA_GRAPH_REPACK", "0") == "1"
        self.fused_lds_pad_kb = int(env("BEVMSDA_FUSED_LDS_PAD", "0"))                                    # co-scheduling probe: occupancy cap of the sampling kernel
        self.stack_free = env("BEVMSDA_STACK_FREE", "1") == "1"      # inference: TSA's [history ; queries] value projected without forming the stack
        self.weight_views = env("BEVMSDA_WEIGHT_VIEWS", "1") == "1"  # training: W^T images packed from W (no transposed copies)
        self.flatten_params = env("BEVMSDA_FLATTEN_PARAMS", "1") == "1"  # training: merged projections' parameters back to back (views, no cat)
        # training: what SCA's forward kernel writes for its backward — 1 = locations and weights (round 5, the default), 2 = the
        # attention weights only (round 6: the backward kernels recompute the locations from the projection rows,
        # bevmsda_backward_rows_offs_*: 564 MB less kept per base step, bit-equal gradients, the same step time — the forward kernel's
        # 32 us come back in the backward kernels' row_src -> offset indirection, profiles/r6/r6t_fused_save_ab.txt), 0 = nothing
        # (a bevmsda_frontend_expand_rows_f32 pass in the backward recomputes both)
        self.fused_save = int(env("BEVMSDA_FUSED_SAVE", "1"))
        self.chain_backward = env("BEVMSDA_CHAIN_BWD", "1") == "1"   # training: the row-local backward of the SCA seam in one kernel
        self.chain_shape = int(env("BEVMSDA_CHAIN_SHAPE", "0"))     # benchmark knob: workgroup shape of the row-chain kernels
        self.grad_thread = env("BEVMSDA_GRAD_THREAD", "1") == "1"   # training: value-projection input gradients summed in the GEMMs
        # autograd path of the encoder layer on the inference kernels (train_ops.py): chain kernels that save what their
        # backward needs, hoisted value projections, device-side row count through forward and backward
        self.train_chain = env("BEVMSDA_TRAIN_CHAIN", "1") == "1"
        self.wgrad_workgroups = int(env("BEVMSDA_WGRAD_WGS", "0"))  # benchmark knob: workgroup target of the multi-problem weight gradient
        self.wgrad_variant = int(env("BEVMSDA_WGRAD_VARIANT", "2"))  # 2 (default, round 6): 256 x 128 tiles on 8 wavefronts with 2 LDS stages where its one-workgroup-per-CU grid fills the chip, else 0; 0: 128 x 128, bf16 planes + transposing LDS reads; 1: gathered fragments
        # training: the parameter-gradient arena of the encoder call being recorded (train_ops.GradArena; set by
        # BEVFormerEncoder.forward for the duration of its call, carried to the backward pass by the Functions' snapshots)
        self.grad_arena = None
        self.use_grad_arena = env("BEVMSDA_GRAD_ARENA", "1") == "1"   # one zero fill for all parameter-gradient accumulators of a pass
        # inference: a layer's last kernel also makes the NEXT layer's TemporalSelfAttention offset / weight projection of
        # the rows it produces (csrc/linear_chain.h TP, BEVFormerEncoder.tsa_seam; round 6)
        self.tsa_seam = env("BEVMSDA_TSA_SEAM", "1") == "1"
        # inference: SpatialCrossAttention's chain kernel walks every camera's row of a slot (idx = q_rows_all) instead of two
        # rows after a stand-alone fold launch (a no-op on most frames, 5 us per layer in a replayed graph)
        self.chain_gather_all = env("BEVMSDA_CHAIN_GATHER_ALL", "1") == "1"
        # inference with overlap_value_proj: the frame-plan kernels are issued on the side stream too, ahead of the camera-value
        # projection (nothing on the main stream reads the plan before the first SpatialCrossAttention joins that stream)
        self.plan_on_side = env("BEVMSDA_PLAN_SIDE", "1") == "1"
        self.overlap_value_proj = env("BEVMSDA_OVERLAP", "1") == "1"  # inference: hoisted SCA value projection on a side stream (its tail rounds and the TSA chain's fill each other: -2 % of the base frame, round 6)
        assert self.gemm in GEMM_MODES, f"BEVMSDA_GEMM must be one of {GEMM_MODES}"

    def snapshot(self):
        return copy.copy(self)


_PROCESS = Modes()
_TLS = threading.local()


def process_defaults():
    """The process-wide defaults (what ``ops.set_*`` edit)."""
    return _PROCESS


def current():
    stack = getattr(_TLS, "stack", None)
    return stack[-1] if stack else _PROCESS


@contextlib.contextmanager
def activate(modes):
    """Make ``modes`` (a snapshot) the calling thread's modes for the block."""
    stack = getattr(_TLS, "stack", None)
    if stack is None:
        stack = _TLS.stack = []
    stack.append(modes)
    try:
        yield modes
    finally:
        stack.pop()


@contextlib.contextmanager
def using(**overrides):
    """``with modes.using(gemm="bf16", value_storage=torch.bfloat16): ...`` — thread-local overrides."""
    m = current().snapshot()
    for k, v in overrides.items():
        if k not in Modes.__slots__:
            raise AttributeError(f"unknown mode {k!r}")
        setattr(m, k, v)
    with activate(m):
        yield m
