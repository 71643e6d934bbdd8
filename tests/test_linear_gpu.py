"""GPU: the MFMA projection kernel (``bevmsda_linear_f32``, csrc/linear_mfma.h) against an
fp64 statement of ``torch.nn.functional.linear``.

Tolerances are error BOUNDS relative to ``|x| @ |w|.T`` (the scale every rounding error of a
dot product is proportional to):
  split (3 bf16 products per fp32 product): 2.5e-5  — per-product error <= 3 * 2^-18 = 1.1e-5
        plus fp32 accumulation; the same bound is asserted for hipBLASLt's fp32 result so the
        test shows both are fp32-class;
  bf16  (operands rounded to bf16):          8e-3   — 2 * 2^-9 per product."""
import pytest
import torch

from bevformer_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BOUND = {"split": 2.5e-5, "bf16": 8e-3}


@pytest.fixture
def gemm_mode():
    saved = ops.gemm_mode()
    yield ops.set_gemm_mode
    ops.set_gemm_mode(saved)
    ops.set_gemm_variant(None, pack=True)


# launch variants of the first kernel (include/bevmsda.h): the fp32 weight matrix split in the kernel (0), the
# pre-split weight image copied by LDS-DMA (12).  The variants that lost (64-deep chunks, dword epilogue, register /
# double-buffered copies, 256-column and 64-row tiles, fragments-first) went with round 2: profiles/r1, tools/experimental
VARIANTS = [None, 0, 12]


def _ref64(x, w, b, relu=False):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    return torch.relu(y) if relu else y


def _scale(x, w):
    return x.abs().double() @ w.abs().double().t() + 1e-30


def _rand(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(DEV)


@pytest.mark.parametrize("variant", VARIANTS[1:])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_variants(gemm_mode, mode, variant):
    """Every launch variant: ragged tile tails in M and N, two sources with an addend, ReLU."""
    gemm_mode(mode)
    ops.set_gemm_variant(variant)
    M, K0, K1, N = 391, 128, 64, 332
    x0, x1, a1 = _rand(M, K0, seed=11), _rand(M, K1, seed=12), _rand(M, K1, seed=13)
    w, b = _rand(N, K0 + K1, seed=14) * 0.1, _rand(N, seed=15)
    with torch.no_grad():
        y = ops.linear(x0, w, b, relu=True, x2=x1, x2_add=a1)
        y2 = ops.linear(x0, w[:, :K0].contiguous(), b)
    xa = torch.cat([x0, x1 + a1], -1)
    err = ((y.double() - _ref64(xa, w, b, relu=True)).abs() / _scale(xa, w)).max().item()
    assert err < BOUND[mode], f"variant {variant}: scaled error {err:.3e}"
    wk = w[:, :K0]
    err2 = ((y2.double() - _ref64(x0, wk, b)).abs() / _scale(x0, wk)).max().item()
    assert err2 < BOUND[mode], f"variant {variant} (single source): scaled error {err2:.3e}"


@pytest.mark.parametrize("mode", ["split", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 256), (1, 5, 32), (129, 131, 64),
                                   (1000, 192, 512), (4099, 768, 256), (257, 64, 96)])
def test_linear_matches_fp64(gemm_mode, mode, M, N, K):
    gemm_mode(mode)
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2) * 0.1, _rand(N, seed=3)
    with torch.no_grad():
        y = ops.linear(x, w, b)
    assert y is not None and y.shape == (M, N)
    err = ((y.double() - _ref64(x, w, b)).abs() / _scale(x, w)).max().item()
    assert err < BOUND[mode], f"{mode} {M}x{N}x{K}: scaled error {err:.3e}"
    if mode == "split":      # hipBLASLt fp32 is in the same class
        lib = ((torch.nn.functional.linear(x, w, b).double() - _ref64(x, w, b)).abs()
               / _scale(x, w)).max().item()
        assert lib < BOUND[mode]


@pytest.mark.parametrize("variant", [None, 0, 12])
def test_linear_identity_with_asymmetric_weight(gemm_mode, variant):
    """A = I picks single weights: catches a transposed / permuted accumulator map, and shows
    that an fp32 weight survives the hi + lo split to 2^-17."""
    gemm_mode("split")
    ops.set_gemm_variant(variant)
    K = 128
    x = torch.eye(K, device=DEV)
    w = torch.arange(256 * K, device=DEV, dtype=torch.float32).reshape(256, K) * 1.0009765625 + 0.3
    with torch.no_grad():
        y = ops.linear(x, w)
    torch.testing.assert_close(y, w.t().contiguous(), rtol=2 ** -16, atol=0)


@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_two_sources_addends_relu_strides(gemm_mode, mode):
    gemm_mode(mode)
    M, K0, K1, N = 777, 256, 256, 192
    big = _rand(M, K0 + 64, seed=4)
    x0 = big[:, 32:32 + K0]                       # row stride K0 + 64, 16-byte aligned offset
    x1, a1 = _rand(1, M, K1, seed=5), _rand(1, M, K1, seed=6)
    a0 = _rand(M, K0, seed=7)
    w, b = _rand(N, K0 + K1, seed=8) * 0.05, _rand(N, seed=9)
    with torch.no_grad():
        y = ops.linear(x0.view(1, M, K0), w, b, relu=True, x_add=a0.view(1, M, K0), x2=x1, x2_add=a1)
    assert y.shape == (1, M, N)
    xa = torch.cat([x0 + a0, (x1 + a1)[0]], -1)
    want = _ref64(xa, w, b, relu=True)
    err = ((y[0].double() - want).abs() / _scale(xa, w)).max().item()
    assert err < BOUND[mode], f"scaled error {err:.3e}"
    assert (y >= 0).all()


def test_linear_not_covered_returns_none(gemm_mode):
    x, w = _rand(10, 48, seed=1), _rand(7, 48, seed=2)        # K not a multiple of 32
    gemm_mode("split")
    with torch.no_grad():
        assert ops.linear(x, w) is None
        torch.testing.assert_close(ops.linear_or_torch(x, w), torch.nn.functional.linear(x, w))
    xg = _rand(10, 64, seed=1).requires_grad_(True)           # autograd -> library GEMM
    assert ops.linear(xg, _rand(7, 64, seed=2)) is None
    gemm_mode("native")
    with torch.no_grad():
        assert ops.linear(_rand(10, 64, seed=1), _rand(7, 64, seed=2)) is None


@pytest.mark.parametrize("variant", [None, 0, 12])
def test_linear_grouped_output(gemm_mode, variant):
    """groups = G: G Linear layers over one input -> (G, M, N / G) contiguous outputs."""
    gemm_mode("split")
    ops.set_gemm_variant(variant)
    M, K, G, n = 300, 64, 3, 256
    x, w, b = _rand(2, M // 2, K, seed=21), _rand(G * n, K, seed=22) * 0.1, _rand(G * n, seed=23)
    with torch.no_grad():
        y = ops.linear(x, w, b, groups=G)
        flat = ops.linear(x, w, b)
    assert y.shape == (G, 2, M // 2, n)
    for g in range(G):
        torch.testing.assert_close(y[g], flat[..., g * n:(g + 1) * n], rtol=0, atol=0)
    with torch.no_grad():
        assert ops.linear(x, w[: G * 192], b[: G * 192], groups=G) is None     # 192 % 128 != 0


def test_packed_weight_is_cached_until_written(gemm_mode):
    gemm_mode("split")
    ops.set_gemm_variant(12)
    x, w = _rand(64, 64, seed=1), _rand(40, 64, seed=2)
    with torch.no_grad():
        y1 = ops.linear(x, w)
        blob = w._bevmsda_pack[1]
        ops.linear(x, w)
        assert w._bevmsda_pack[1] is blob
        w.mul_(2.0)                                   # in-place write bumps the version
        y2 = ops.linear(x, w)
        assert w._bevmsda_pack[1] is not blob
    torch.testing.assert_close(y2, 2 * y1, rtol=1e-5, atol=1e-5)


def test_linear_propagates_nan_rows_only(gemm_mode):
    gemm_mode("split")
    x, w = _rand(256, 64, seed=1), _rand(128, 64, seed=2)
    x[17, 5] = float("nan")
    with torch.no_grad():
        y = ops.linear(x, w, relu=True)
    assert torch.isnan(y[17]).all()
    assert torch.isfinite(torch.cat([y[:17], y[18:]])).all()


@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_gather_mean_is_the_two_step_result(gemm_mode, mode):
    """Camera mean folded into the A-load: bit-identical to gather_mean followed by linear."""
    gemm_mode(mode)
    g = torch.Generator().manual_seed(31)
    R, Q, C = 700, 517, 256
    rows = _rand(R, C, seed=32)
    idx = torch.randint(0, R, (Q, 2), generator=g, dtype=torch.int32)
    idx[torch.rand(Q, generator=g) < 0.7, 1] = -1            # most queries: one camera
    idx[:5] = -1                                             # a few: none
    cnt = (idx >= 0).sum(1).clamp(min=1).float()
    scale, w, b = (1.0 / cnt).to(DEV), _rand(256, C, seed=33) * 0.1, _rand(256, seed=34)
    idx = idx.to(DEV)
    with torch.no_grad():
        got = ops.linear_gather_mean(rows, idx, scale, w, b)
        want = ops.linear(ops.gather_mean(rows, idx, scale), w, b)
    assert got is not None and torch.equal(got, want)
    assert torch.equal(got[:5], b.expand(5, -1))             # empty rows: bias only


@pytest.mark.parametrize("variant", [None, 0, 12])
def test_linear_bf16_output_is_the_rounded_fp32_result(gemm_mode, variant):
    gemm_mode("split")
    ops.set_gemm_variant(variant)
    x, w, b = _rand(3, 100, 64, seed=41), _rand(512, 64, seed=42) * 0.1, _rand(512, seed=43)
    with torch.no_grad():
        y32 = ops.linear(x, w, b, groups=2)
        y16 = ops.linear(x, w, b, groups=2, out_dtype=torch.bfloat16)
    assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape
    assert torch.equal(y16, y32.to(torch.bfloat16))


@pytest.mark.parametrize("relu", [False, True])
def test_linear_autograd_function_matches_torch(gemm_mode, relu):
    """Under autograd ``linear_or_torch`` is an autograd Function: input gradient on the MFMA
    kernel (over the transposed weight), forward on the library GEMM or (knob) on the kernel."""
    gemm_mode("split")
    x = _rand(3, 70, 128, seed=51).requires_grad_(True)
    w = (_rand(96, 128, seed=52) * 0.1).requires_grad_(True)
    b = _rand(96, seed=53).requires_grad_(True)
    g = _rand(3, 70, 96, seed=54)
    y = ops.linear_or_torch(x, w, b, relu=relu)
    assert y.grad_fn is not None and "LinearFunction" in type(y.grad_fn).__name__
    y.backward(g)
    xr, wr, br = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    if relu:
        yr = torch.relu(yr)
    yr.backward(g)
    torch.testing.assert_close(y.detach(), yr.detach(), rtol=1e-4, atol=1e-4)
    for got, want, what in ((x.grad, xr.grad, "dx"), (w.grad, wr.grad, "dw"), (b.grad, br.grad, "db")):
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        assert err < 1e-4, f"{what}: {err:.2e}"


@pytest.mark.parametrize("M,K", [(1000, 256), (4099, 512), (128, 256)])
@pytest.mark.parametrize("with_res", [True, False])
def test_linear_layernorm_fused_epilogue(M, K, with_res):
    """``bevmsda_linear_panel_f32`` with a LayerNorm descriptor: LayerNorm(x W^T + b + res) against the fp64 statement of
    the three torch ops; ragged last row tile, K = 256 / 512, with and without residual."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(256, K, generator=g) * K ** -0.5
    b = torch.randn(256, generator=g) * 0.1
    res = torch.randn(M, 256, generator=g) if with_res else None
    norm = torch.nn.LayerNorm(256)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(256, generator=g) * 0.2 + 1.0)
        norm.bias.copy_(torch.randn(256, generator=g) * 0.1)
    y64 = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if with_res:
        y64 = y64 + res.double()
    want = torch.nn.functional.layer_norm(y64, (256,), norm.weight.double(), norm.bias.double(), norm.eps)
    norm = norm.to(DEV)
    with torch.no_grad(), ops.using(ln_fuse=True):
        got = ops.linear_layernorm(x.to(DEV), w.to(DEV), b.to(DEV), res.to(DEV) if with_res else None, norm)
    assert got is not None and got.shape == (M, 256)
    # (LayerNorm divides by the row's standard deviation: the split-bf16 product round-off of 4e-6 x
    # |x||w| shows up as ~3e-5 on rows of unit-scale output)
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-4, atol=1e-4)


def test_linear_layernorm_with_camera_gather():
    """The SCA form: A = scale * (rows[idx0] + rows[idx1]) gathered in the A-load, then projection +
    residual + LayerNorm."""
    g = torch.Generator().manual_seed(5)
    R, Q = 700, 500
    rows = torch.randn(R, 256, generator=g)
    idx = torch.full((Q, 2), -1, dtype=torch.int32)
    perm = torch.randperm(R, generator=g)
    idx[:, 0] = perm[:Q].int()
    idx[:200, 1] = perm[Q:Q + 200].int()
    idx[450:, 0] = -1                                     # queries no camera sees
    scale = 1.0 / (idx >= 0).sum(1).clamp(min=1).float()
    w = torch.randn(256, 256, generator=g) / 16
    b = torch.randn(256, generator=g) * 0.1
    res = torch.randn(Q, 256, generator=g)
    norm = torch.nn.LayerNorm(256)
    a = torch.zeros(Q, 256, dtype=torch.float64)
    for j in range(2):
        ok = idx[:, j] >= 0
        a[ok] += rows[idx[ok, j].long()].double()
    a = a * scale[:, None].double()
    want = torch.nn.functional.layer_norm(torch.nn.functional.linear(a, w.double(), b.double()) + res.double(), (256,),
                                          norm.weight.double(), norm.bias.double(), norm.eps)
    norm = norm.to(DEV)
    with torch.no_grad(), ops.using(ln_fuse=True):
        got = ops.linear_layernorm(rows.to(DEV), w.to(DEV), b.to(DEV), res.to(DEV), norm,
                                   gather=(idx.to(DEV), scale.to(DEV)))
    assert got is not None
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (4099, 192, 512), (130, 768, 256), (40000, 256, 256), (33, 64, 32)])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_wgrad_kernel(M, N, K, mode):
    """csrc/wgrad_mfma.h: grad_W = g^T x and grad_b = g.sum(0) against their fp64 statements — ragged row
    slices, N not a multiple of the 128-column tile, tiny and base-size M."""
    g_ = torch.Generator().manual_seed(M + N + K)
    g = torch.randn(M, N, generator=g_)
    x = torch.randn(M, K, generator=g_)
    want_w = g.double().t() @ x.double()
    want_b = g.double().sum(0)
    saved = ops.gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        gw, gb = ops.linear_wgrad(g.to(DEV), x.to(DEV), True)
    finally:
        ops.set_gemm_mode(saved)
    assert gw is not None and gw.shape == (N, K) and gb.shape == (N,)
    scale = (g.double().abs().t() @ x.double().abs())
    tol = 2.5e-5 if mode == "split" else 8e-3
    assert ((gw.cpu().double() - want_w).abs() / scale.clamp(min=1e-9)).max().item() < tol * 40
    assert ((gw.cpu().double() - want_w).abs().max() / scale.max()).item() < tol
    torch.testing.assert_close(gb.cpu().double(), want_b, rtol=1e-4, atol=1e-4 * M ** 0.5)


@pytest.mark.parametrize("M,K0,K1,N,relu,groups,out", [
    (1000, 256, 0, 256, False, 1, torch.float32), (4099, 256, 256, 192, False, 1, torch.float32),
    (513, 256, 0, 512, True, 1, torch.float32), (2050, 256, 0, 768, False, 3, torch.float32),
    (300, 512, 0, 256, False, 1, torch.float32), (1111, 256, 0, 256, False, 2, torch.bfloat16)])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_software_pipelined_kernel(M, K0, K1, N, relu, groups, out, mode):
    """csrc/linear_pipe.h (two LDS stages, two fragment register sets, activations two chunks ahead, one
    barrier per chunk; K loop unrolled for K = 256 and 512) against the first kernel: same products in the
    same order, so bit-identical — ragged row tiles, two K sources, ReLU, grouped and bf16 output."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K0, generator=g).to(DEV)
    x2 = torch.randn(M, K1, generator=g).to(DEV) if K1 else None
    w = (torch.randn(N, K0 + K1, generator=g) * (K0 + K1) ** -0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    saved = ops.gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        res = {}
        for kern in ("pipe", "first"):
            ops.set_gemm_kernel(kern)
            with torch.no_grad():
                y = ops.linear(x, w, b, relu=relu, x2=x2, groups=groups, out_dtype=out)
            assert y is not None
            res[kern] = y.float().cpu()
    finally:
        ops.set_gemm_kernel(None)
        ops.set_gemm_mode(saved)
    assert torch.equal(res["pipe"], res["first"])


@pytest.mark.parametrize("M,K0,K1,N,relu,groups,out", [
    (1000, 256, 0, 256, False, 1, torch.float32), (4099, 256, 256, 192, False, 1, torch.float32),
    (777, 256, 0, 512, True, 1, torch.float32), (130, 512, 0, 256, False, 1, torch.float32),
    (3001, 256, 0, 1536, False, 6, torch.float32), (3001, 256, 0, 1536, False, 6, torch.bfloat16),
    (5000, 256, 0, 768, False, 1, torch.float32), (1, 256, 0, 64, False, 1, torch.float32),
    (63, 256, 0, 100, True, 1, torch.float32), (40000, 256, 0, 256, False, 1, torch.float32)])
@pytest.mark.parametrize("kernel", ["panel64", "panel128"])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_row_panel_kernel(gemm_mode, mode, kernel, M, K0, K1, N, relu, groups, out):
    """``bevmsda_linear_panel_f32`` (csrc/linear_panel.h), both panel shapes: against the fp64 statement and against
    the first kernel (same arithmetic, another k order inside the MFMAs: fp32 summation-order differences only)."""
    gemm_mode(mode)
    x0 = _rand(M, K0, seed=61)
    x1 = _rand(M, K1, seed=62) if K1 else None
    a1 = _rand(M, K1, seed=63) if K1 else None
    a0 = _rand(M, K0, seed=64) if K1 else None          # addends on both sources in the two-source case
    w, b = _rand(N, K0 + K1, seed=65) * 0.05, _rand(N, seed=66)
    try:
        with torch.no_grad():
            ops.set_gemm_kernel(kernel)
            called = []
            ops.set_gemm_timer(lambda tag, f, n: (called.append(tag), ops._NoTimer())[1])
            y = ops.linear(x0, w, b, relu=relu, x_add=a0, x2=x1, x2_add=a1, groups=groups, out_dtype=out, tag="panel")
            ops.set_gemm_timer(None)
            ops.set_gemm_kernel("first")
            y1 = ops.linear(x0, w, b, relu=relu, x_add=a0, x2=x1, x2_add=a1, groups=groups, out_dtype=out)
    finally:
        ops.set_gemm_kernel(None)
        ops.set_gemm_timer(None)
    assert y is not None and called == ["panel"]
    xa = x0 if not K1 else torch.cat([x0 + a0, x1 + a1], -1)
    want = _ref64(xa, w, b, relu=relu)
    if groups > 1:
        want = want.view(M, groups, N // groups).transpose(0, 1)
    bound = BOUND[mode] if out == torch.float32 else 8e-3
    sc = _scale(xa, w) + (b.abs().double() if out != torch.float32 else 0)
    if groups > 1:
        sc = sc.view(M, groups, N // groups).transpose(0, 1)
    err = ((y.double() - want).abs() / sc).max().item()
    assert err < bound, f"{kernel} {mode}: scaled error {err:.3e}"
    if out == torch.float32:
        d = ((y.double() - y1.double()).abs() / sc).max().item()
        assert d < (2e-6 if mode == "split" else 1e-6), f"{kernel} vs the first kernel: {d:.3e}"


@pytest.mark.parametrize("seg_len,nseg,empty", [(375, 12, (0, 3, 4, 11)), (64, 9, (1, 2, 3)), (50, 7, ()), (1000, 3, (0, 1, 2)),
                                                (37, 40, tuple(range(5, 31)))])
@pytest.mark.parametrize("halo_w", [None, 0, 29])
@pytest.mark.parametrize("kernel", ["panel64", "panel128"])
def test_linear_row_panel_skips_unused_row_segments(gemm_mode, kernel, seg_len, nseg, empty, halo_w):
    """``bevmsda_linear_panel_segments_f32``: rows of segments WITH entries (device-side starts) are exactly those of
    the full launch; a workgroup returns early only when all of its rows lie in empty segments, so rows of empty
    segments are either untouched (the NaN prefill) or equal to the full result — never anything else.  With level
    shapes the rows within max W + 1 of a used segment count as used (the sampling kernels' zero-weight taps)."""
    gemm_mode("split")
    M, K, N, L = seg_len * nseg, 256, 768, 3
    x, w, b = _rand(M, K, seed=71), _rand(N, K, seed=72) * 0.05, _rand(N, seed=73)
    counts = torch.tensor([0 if i in empty else 1 + (i * 7) % 5 for i in range(nseg)])
    start = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).to(torch.int32).to(DEV)
    ops._SEGMENT_POISON["on"] = True
    try:
        with torch.no_grad(), ops.using(gemm_kernel=kernel):
            full = ops.linear(x, w, b, groups=L)
            shapes = None if halo_w is None else torch.tensor([[3, halo_w], [2, halo_w // 2]], device=DEV)
            part = ops.linear(x, w, b, groups=L, segments=(start, seg_len) + ((shapes,) if shapes is not None else ()))
    finally:
        ops._SEGMENT_POISON["on"] = False
    assert torch.isfinite(full).all()
    full, part = full.view(L, nseg, seg_len, -1), part.view(L, nseg, seg_len, -1)
    untouched = 0
    for i in range(nseg):
        if i in empty:
            row_nan = torch.isnan(part[:, i]).all(-1).all(0)             # (seg_len,) rows left alone in every group
            row_same = (part[:, i] == full[:, i]).all(-1).all(0)
            assert (row_nan | row_same).all()
            untouched += int(row_nan.sum())
        else:
            assert torch.equal(part[:, i], full[:, i])
    bm = 64 if kernel == "panel64" else 128
    # every aligned run of bm rows that lies wholly in empty segments must have been skipped
    rows_empty = torch.ones(M, dtype=torch.bool)
    halo = 0 if halo_w is None else halo_w + 1
    for i in range(nseg):
        if i not in empty:
            rows_empty[max(0, i * seg_len - halo):(i + 1) * seg_len + halo] = False
    want = sum(int(min(m0 + bm, M) - m0) for m0 in range(0, M, bm) if rows_empty[m0:m0 + bm].all())
    assert untouched == want


@pytest.mark.parametrize("n,k,col0", [(256, 256, 0), (256, 768, 0), (512, 192, 0), (200, 96, 0), (256, 192, 256)])
def test_weight_image_packed_from_the_transposed_matrix(gemm_mode, n, k, col0):
    """``bevmsda_linear_pack_weight_t_f32``: the image of W^T built from W where it lies (a transposed view, also of a
    column block of a wider matrix — the two halves of TSA's two-source weight) is the image of the contiguous
    transpose, byte for byte; ``ops.linear`` over the view equals ``ops.linear`` over the copy."""
    gemm_mode("split")
    wide = _rand(k, col0 + n + 8, seed=91)          # W: (k = out features, in features ...), row-major
    view = wide[:, col0:col0 + n].t()                # (n, k) view of W[:, col0 : col0 + n]^T
    assert ops._is_transposed_view(view)
    blob_t = ops.packed_weight(view)
    blob_c = ops.packed_weight(view.contiguous())
    assert blob_t is not None and torch.equal(blob_t, blob_c)
    x = _rand(777, k, seed=92)
    with torch.no_grad(), ops.using(gemm_kernel="first"):
        got = ops.linear(x, view, None, _inside_autograd=True)
        want = ops.linear(x, view.contiguous(), None, _inside_autograd=True)
    assert got is not None and torch.equal(got, want)


@pytest.mark.parametrize("m0,m1", [(4000, 4000), (1000, 37), (33, 5000), (64, 64), (1, 1)])
@pytest.mark.parametrize("kernel,out", [(None, torch.float32), ("panel64", torch.bfloat16), ("panel128", torch.float32)])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_linear_row_panel_two_row_blocks(gemm_mode, m0, m1, kernel, out, mode):
    """``bevmsda_linear_panel_rows2_f32`` (TSA's value [history ; queries] projected without the stack): the same
    kernel over the same rows — equal, bit for bit, to the launch over the concatenated tensor; a split inside a panel
    (m0 not a multiple of 64 / 128) included."""
    gemm_mode(mode)
    K, N, L = 256, 1536, 6
    lo, hi = _rand(m0, K, seed=81), _rand(m1, K, seed=82)
    w, b = _rand(N, K, seed=83) * 0.05, _rand(N, seed=84)
    with torch.no_grad(), ops.using(gemm_kernel=kernel):
        got = ops.linear_rows2(lo, hi, w, b, groups=L, out_dtype=out, tag="t")
        want = ops.linear(torch.cat([lo, hi], 0), w, b, groups=L, out_dtype=out)
    assert got is not None and got.shape == (L, m0 + m1, N // L) and got.dtype == out
    assert torch.equal(got, want)
    ref = (torch.cat([lo, hi], 0).double() @ w.double().t() + b.double()).view(m0 + m1, L, N // L).transpose(0, 1)
    assert (got.double() - ref).abs().max().item() < (5e-2 if out == torch.bfloat16 or mode == "bf16" else 1e-4)


@pytest.mark.parametrize("kernel", ["panel64", "panel128"])
def test_linear_row_panel_identity_with_asymmetric_weight(gemm_mode, kernel):
    """A = I picks single weights: catches a permuted k order between the activation image and the weight image,
    or a transposed accumulator map."""
    gemm_mode("split")
    K = 256
    x = torch.eye(K, device=DEV)
    w = torch.arange(320 * K, device=DEV, dtype=torch.float32).reshape(320, K) * 1.0009765625 + 0.3
    try:
        ops.set_gemm_kernel(kernel)
        with torch.no_grad():
            y = ops.linear(x, w)
    finally:
        ops.set_gemm_kernel(None)
    torch.testing.assert_close(y, w.t().contiguous(), rtol=2 ** -16, atol=0)


@pytest.mark.parametrize("kernel", ["panel64", "panel128"])
def test_linear_row_panel_layernorm_and_gather(gemm_mode, kernel):
    """LayerNorm epilogue (K = 256 and 512) and the camera gather of the row-panel kernel against the unfused
    sequence on the same kernel."""
    gemm_mode("split")
    g = torch.Generator().manual_seed(7)
    R, Q = 900, 641
    rows = _rand(R, 256, seed=71)
    idx = torch.randint(0, R, (Q, 2), generator=g, dtype=torch.int32)
    idx[torch.rand(Q, generator=g) < 0.6, 1] = -1
    idx[:7] = -1
    idx[7:9, 0] = -1                                           # only the second slot filled
    scale = (1.0 / (idx >= 0).sum(1).clamp(min=1).float()).to(DEV)
    idx = idx.to(DEV)
    w, b, res = _rand(256, 256, seed=72) / 16, _rand(256, seed=73) * 0.1, _rand(Q, 256, seed=74)
    w5, x5 = _rand(256, 512, seed=75) / 22, _rand(Q, 512, seed=76)
    norm = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(_rand(256, seed=77) * 0.2 + 1.0)
        norm.bias.copy_(_rand(256, seed=78) * 0.1)
    with torch.no_grad(), ops.using(gemm_kernel=kernel, ln_fuse=True):
        got = ops.linear_layernorm(rows, w, b, res, norm, gather=(idx, scale))
        two = ops.linear_gather_mean(rows, idx, scale, w, b)
        three = ops.linear(ops.gather_mean(rows, idx, scale), w, b)
        got5 = ops.linear_layernorm(x5, w5, b, res, norm)
        lin5 = ops.linear(x5, w5, b)
    assert got is not None and two is not None and got5 is not None
    assert torch.equal(two, three)                             # gather in the split pass == gather_mean + projection
    want = torch.nn.functional.layer_norm(two.double() + res.double(), (256,), norm.weight.double(), norm.bias.double(), norm.eps)
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5)
    want5 = torch.nn.functional.layer_norm(lin5.double() + res.double(), (256,), norm.weight.double(), norm.bias.double(), norm.eps)
    torch.testing.assert_close(got5.double(), want5, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,M,with_gather", [(sh, M, wg) for sh in (1, 2) for M, wg in ((641, True), (4099, False), (64, True), (1, False))]
                         + [(0, 40000, True)])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_proj_ffn_chain_kernel_is_the_three_launch_sequence(gemm_mode, mode, M, with_gather, shape):
    """``bevmsda_proj_ffn_chain_f32`` (csrc/linear_chain.h): output projection (+ camera gather) + residual + LayerNorm
    + FFN + residual + LayerNorm in one kernel against the same chain as three launches of the row-panel kernel (the
    intermediate x and hidden activations are split from the same fp32 values: differences come from the order of the
    LayerNorm statistics only) and against the fp64 statement.  Shapes 1 / 2: 64- / 32-row panels; 0: the library's
    choice by row count (here at the base row count)."""
    gemm_mode(mode)
    g = torch.Generator().manual_seed(M)
    R = max(M + 37, 100)
    rows = _rand(R, 256, seed=81)
    idx = scale = None
    if with_gather:
        idx = torch.randint(0, R, (M, 2), generator=g, dtype=torch.int32)
        idx[torch.rand(M, generator=g) < 0.6, 1] = -1
        idx[:3] = -1
        scale = (1.0 / (idx >= 0).sum(1).clamp(min=1).float()).to(DEV)
        idx = idx.to(DEV)
    w0, b0, res = _rand(256, 256, seed=82) / 16, _rand(256, seed=83) * 0.1, _rand(M, 256, seed=84)
    fc1, fc2 = torch.nn.Linear(256, 512).to(DEV), torch.nn.Linear(512, 256).to(DEV)
    n0, n1 = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        for n, s in ((n0, 85), (n1, 87)):
            n.weight.copy_(_rand(256, seed=s) * 0.2 + 1.0)
            n.bias.copy_(_rand(256, seed=s + 1) * 0.1)
        gather = (idx, scale) if with_gather else None
        src = rows if with_gather else rows[:M]
        with ops.using(ln_fuse=True, gemm_kernel="panel64", chain_shape=shape):
            got = ops.proj_ffn_chain(src, w0, b0, res, n0, fc1, fc2, n1, gather=gather)
            x = ops.linear_layernorm(src, w0, b0, res, n0, gather=gather)
            h = ops.linear(x, fc1.weight, fc1.bias, relu=True)
            want = ops.linear_layernorm(h, fc2.weight, fc2.bias, x, n1)
        assert got is not None and x is not None and want is not None and got.shape == (M, 256)
        # (bf16 operands: a last-bit difference of x flips its bf16 rounding for single elements)
        t3 = 2e-5 if mode == "split" else 2e-2
        torch.testing.assert_close(got, want, rtol=t3, atol=t3)
        # fp64 statement
        a = src.double() if not with_gather else sum(
            torch.where((idx[:, j] >= 0)[:, None], rows.double()[idx[:, j].clamp(min=0).long()], torch.zeros(1, dtype=torch.float64, device=DEV))
            for j in range(2)) * scale.double()[:, None]
        ln = torch.nn.functional.layer_norm
        x64 = ln(a @ w0.double().t() + b0.double() + res.double(), (256,), n0.weight.double(), n0.bias.double(), n0.eps)
        y64 = ln(x64 + torch.relu(x64 @ fc1.weight.double().t() + fc1.bias.double()) @ fc2.weight.double().t() + fc2.bias.double(),
                 (256,), n1.weight.double(), n1.bias.double(), n1.eps)
    tol = 2e-4 if mode == "split" else 5e-2
    torch.testing.assert_close(got.double(), y64, rtol=tol, atol=tol)


@pytest.mark.parametrize("shape,M,with_gather,with_pos", [(sh, M, wg, wp) for sh in (1, 2)
                                                         for M, wg, wp in ((641, True, True), (4099, False, True), (64, True, False), (1, False, True))]
                         + [(0, 40000, True, True), (0, 16500, False, True)])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_proj_ffn_chain_tail_is_the_chain_plus_the_two_source_projection(gemm_mode, mode, M, with_gather, with_pos, shape):
    """``bevmsda_proj_ffn_chain_tail_f32`` (csrc/linear_chain.h TP): the chain kernel that also forms the NEXT layer's
    TemporalSelfAttention projection ``[first | y + pos] W3^T + b3`` of the rows it produces — y bit-equal to the plain chain
    launch, the projection against the stand-alone two-source launch on that y (same split operands: fp32 summation order
    only) and against the fp64 statement.  Row counts with a partial last panel, one row, the base grid (mixed 64- / 32-row
    launches) and a count whose tail launch starts at a row offset."""
    gemm_mode(mode)
    g = torch.Generator().manual_seed(M + 5)
    R = max(M + 37, 100)
    rows = _rand(R, 256, seed=181)
    idx = scale = None
    if with_gather:
        idx = torch.randint(0, R, (M, 2), generator=g, dtype=torch.int32)
        idx[torch.rand(M, generator=g) < 0.6, 1] = -1
        scale = (1.0 / (idx >= 0).sum(1).clamp(min=1).float()).to(DEV)
        idx = idx.to(DEV)
    w0, b0, res = _rand(256, 256, seed=182) / 16, _rand(256, seed=183) * 0.1, _rand(M, 256, seed=184)
    fc1, fc2 = torch.nn.Linear(256, 512).to(DEV), torch.nn.Linear(512, 256).to(DEV)
    n0, n1 = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    first = _rand(1, M, 256, seed=185)
    pos = _rand(1, M, 256, seed=186) if with_pos else None
    w3, b3 = _rand(192, 512, seed=187) / 16, _rand(192, seed=188) * 0.1
    with torch.no_grad():
        for n, sd in ((n0, 85), (n1, 87)):
            n.weight.copy_(_rand(256, seed=sd) * 0.2 + 1.0)
            n.bias.copy_(_rand(256, seed=sd + 1) * 0.1)
        gather = (idx, scale) if with_gather else None
        src = rows if with_gather else rows[:M]
        # (left to the library the tail launch takes 32-row workgroups at every row count; the plain chain's LayerNorm sums
        # its statistics in another order on 64-row workgroups: bit-equality is per shape)
        with ops.using(ln_fuse=True, chain_shape=shape or 2):
            plain = ops.proj_ffn_chain(src, w0, b0, res, n0, fc1, fc2, n1, gather=gather)
        with ops.using(ln_fuse=True, chain_shape=shape):
            got = ops.proj_ffn_chain(src, w0, b0, res, n0, fc1, fc2, n1, gather=gather, tail=(first, pos, w3, b3))
        assert plain is not None and got is not None and got[1] is not None
        y, pr = got
        assert torch.equal(y, plain) and pr.shape == (M, 192)
        with ops.using(gemm_kernel="first"):
            two = ops.linear(first, w3, b3, x2=y.view(1, M, 256), x2_add=pos)
        assert two is not None
        t2 = 5e-5 if mode == "split" else 2e-2
        torch.testing.assert_close(pr, two.view(M, 192), rtol=t2, atol=t2)
        q64 = y.double() + (pos.double().view(M, 256) if with_pos else 0.0)
        p64 = torch.cat([first.double().view(M, 256), q64], -1) @ w3.double().t() + b3.double()
    tol = 2e-4 if mode == "split" else 5e-2
    torch.testing.assert_close(pr.double(), p64, rtol=tol, atol=tol)
    # a weight the kernel does not tile (N3 = 96) is dropped, the chain itself still runs
    with torch.no_grad(), ops.using(ln_fuse=True, chain_shape=shape or 2):
        fb = ops.proj_ffn_chain(src, w0, b0, res, n0, fc1, fc2, n1, gather=gather, tail=(first, pos, w3[:96].contiguous(), b3[:96].contiguous()))
    assert fb is not None and fb[1] is None and torch.equal(fb[0], plain)


@pytest.mark.parametrize("shape,M,N2", [(sh, M, n2) for sh in (1, 2) for M, n2 in ((641, 768), (4099, 192), (64, 96), (5000, 768))]
                         + [(0, 40000, 768)])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_proj_ln_proj_chain_kernel_is_the_two_launch_sequence(gemm_mode, mode, M, N2, shape):
    """``bevmsda_proj_ln_proj_chain_f32`` (csrc/linear_chain.h, MODE 1): output projection + residual + LayerNorm (stored)
    + the next attention's projection of the normed rows, against the LayerNorm-fused projection followed by a plain
    projection, and against the fp64 statement."""
    gemm_mode(mode)
    rows, res = _rand(M, 256, seed=91), _rand(M, 256, seed=92)
    w0, b0 = _rand(256, 256, seed=93) / 16, _rand(256, seed=94) * 0.1
    w1, b1 = _rand(N2, 256, seed=95) / 16, _rand(N2, seed=96) * 0.1
    n0 = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        n0.weight.copy_(_rand(256, seed=97) * 0.2 + 1.0)
        n0.bias.copy_(_rand(256, seed=98) * 0.1)
        with ops.using(ln_fuse=True, chain_shape=shape):
            got = ops.proj_ln_proj_chain(rows, w0, b0, res, n0, w1, b1)
            x = ops.linear_layernorm(rows, w0, b0, res, n0)
        with ops.using(gemm_kernel="first"):
            p = ops.linear(x, w1, b1)
        assert got is not None and x is not None
        gx, gp = got
        assert gx.shape == (M, 256) and gp.shape == (M, N2)
        t2 = 2e-5 if mode == "split" else 2e-2
        torch.testing.assert_close(gx, x, rtol=t2, atol=t2)
        torch.testing.assert_close(gp, p, rtol=max(t2, 5e-5), atol=max(t2, 5e-5))
        ln = torch.nn.functional.layer_norm
        x64 = ln(rows.double() @ w0.double().t() + b0.double() + res.double(), (256,), n0.weight.double(), n0.bias.double(), n0.eps)
        p64 = x64 @ w1.double().t() + b1.double()
    tol = 2e-4 if mode == "split" else 5e-2
    torch.testing.assert_close(gx.double(), x64, rtol=tol, atol=tol)
    torch.testing.assert_close(gp.double(), p64, rtol=tol, atol=tol)


@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_shared_input_gradient_is_the_sum_autograd_would_form(gemm_mode, mode):
    """``ops.GradThread``: several projections of one tensor sum their input gradients inside the GEMM epilogue
    (``bevmsda_linear_desc.reserved[0] = 1``) instead of through autograd's adds — same gradients as without it (to the
    order of the fp32 additions), for the shared input, for every weight / bias, on a second backward through a retained
    graph too, when some consumers' outputs do not reach the loss, and when the tensor has other consumers as well."""
    gemm_mode(mode)
    M, K = 3001, 256
    ws = [(_rand(256, K, seed=101 + i) / 16).requires_grad_(True) for i in range(4)]
    bs = [(_rand(256, seed=111 + i) * 0.1).requires_grad_(True) for i in range(4)]
    x = _rand(7, M // 7 + 1, K, seed=121)[:, :M // 7].contiguous()            # a leading shape that needs reshaping
    coef = [_rand(*x.shape[:-1], 256, seed=131 + i) for i in range(4)]

    def run(share, used=(0, 1, 2, 3)):
        xi = x.clone().requires_grad_(True)
        sh = ops.GradThread() if share else None
        ys = [ops.linear_or_torch(xi, ws[i], bs[i], tag=f"p{i}", thread=sh) for i in range(4)]
        loss = sum((ys[i] * coef[i]).sum() for i in used) + (xi[:3] * coef[0][:3]).sum() * 0.5      # + a consumer outside the thread
        params = [xi] + [ws[i] for i in used] + [bs[i] for i in used]
        g1 = torch.autograd.grad(loss, params, retain_graph=True)
        g2 = torch.autograd.grad(loss, params)
        return g1, g2

    want, _ = run(False)
    got1, got2 = run(True)
    tol = 2e-5 if mode == "split" else 1e-4
    for a_, b_, c_ in zip(want, got1, got2):
        scale = a_.abs().max().item()
        assert (a_ - b_).abs().max().item() <= tol * scale and (a_ - c_).abs().max().item() <= tol * scale
    # the adds are gone: the shared run issues accumulate launches (same launch count, no elementwise sum) — and a
    # consumer that takes no part in the loss must not leave the accumulator half-finished for the next use
    want3, _ = run(False, used=(0, 2))
    got3, got3b = run(True, used=(0, 2))
    for a_, b_, c_ in zip(want3, got3, got3b):
        scale = a_.abs().max().item()
        assert (a_ - b_).abs().max().item() <= tol * scale and (a_ - c_).abs().max().item() <= tol * scale
