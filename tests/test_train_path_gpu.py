"""GPU: the autograd path of the encoder layer on the inference kernels (bevformer_amd/train_ops.py).

Each autograd Function against autograd through the float64 torch statement of the same math (the statements are the
reference's: temporal_self_attention.py:186-211, 267-272; spatial_cross_attention.py:165-175; encoder.py:376-404);
then the whole encoder: fast path == per-op autograd path (``train_chain`` off) on outputs and every gradient, the
device-side row count through forward and backward, and a HIP graph of a complete forward + backward step."""
import pytest
import torch
import torch.nn as nn

from bevformer_amd import ops, train_ops
from bevformer_amd import synthetic as S

from helpers import build_pair

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _check(got, want, what, l2=2e-4, mx=5e-4):
    e2, em = _rel(got, want)
    assert e2 < l2 and em < mx, f"{what}: relative L2 {e2:.2e} (< {l2}), max error / max {em:.2e} (< {mx})"


def _leaf(t):
    return t.clone().detach().to(DEV).requires_grad_(True)


def _ln(x, g, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, eps)


@pytest.mark.parametrize("M,N2", [(200, 768), (9001, 768), (3000, 192)])
def test_seam_t_function_matches_the_torch_statement(M, N2):
    g = torch.Generator().manual_seed(M)
    rows, res = torch.randn(M, 256, generator=g), torch.randn(1, M, 256, generator=g)
    w0, b0 = torch.randn(256, 256, generator=g) * 0.06, torch.randn(256, generator=g) * 0.1
    w1, b1 = torch.randn(N2, 256, generator=g) * 0.06, torch.randn(N2, generator=g) * 0.1
    norm = nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
        norm.bias.copy_(0.1 * torch.randn(256, generator=g))
    gx, gp = torch.randn(1, M, 256, generator=g), torch.randn(M, N2, generator=g)
    leaves = [_leaf(t) for t in (rows, w0, b0, res, w1, b1)]
    out = train_ops.seam_t(leaves[0], leaves[1], leaves[2], leaves[3], norm, leaves[4], leaves[5])
    assert out is not None
    x, p = out
    torch.autograd.backward([x, p], [gx.to(DEV), gp.to(DEV)])
    # float64 statement
    d = [t.double().clone().requires_grad_(True) for t in (rows, w0, b0, res, w1, b1)]
    gam, bet = norm.weight.detach().double().cpu().requires_grad_(True), norm.bias.detach().double().cpu().requires_grad_(True)
    xw = _ln(d[0] @ d[1].t() + d[2] + d[3], gam, bet, norm.eps)
    pw = xw.reshape(M, 256) @ d[4].t() + d[5]
    torch.autograd.backward([xw, pw], [gx.double(), gp.double()])
    _check(x, xw, "x", 1e-5, 1e-4)
    _check(p, pw, "p", 1e-5, 1e-4)
    for name, a, b_ in zip(("rows", "w0", "b0", "res", "w1", "b1"), leaves, d):
        _check(a.grad, b_.grad, "grad " + name)
    _check(norm.weight.grad, gam.grad, "grad gamma")
    _check(norm.bias.grad, bet.grad, "grad beta")


def _seam_s_case(M, R, seed, nan_tail=0):
    """rows (R + nan_tail, 256) of which R are real; every query has 0..2 rows (idx), scale = 1 / count."""
    g = torch.Generator().manual_seed(seed)
    rows = torch.randn(R, 256, generator=g)
    row_slot = torch.randint(0, M, (R,), generator=g)
    # at most two rows per slot: drop surplus rows by re-assigning them to empty slots
    idx = torch.full((M, 2), -1, dtype=torch.int32)
    fill = torch.zeros(M, dtype=torch.long)
    free = [m for m in range(M)]
    for r in range(R):
        s = int(row_slot[r])
        while fill[s] >= 2:
            s = (s + 1) % M
        row_slot[r] = s
        idx[s, fill[s]] = r
        fill[s] += 1
    scale = 1.0 / fill.clamp(min=1).float()
    if nan_tail:
        rows = torch.cat([rows, torch.full((nan_tail, 256), float("nan"))], 0)
        row_slot = torch.cat([row_slot, torch.full((nan_tail,), 2 ** 30, dtype=row_slot.dtype)])     # garbage beyond the count
    return rows, idx, scale, row_slot.to(torch.int32), g


@pytest.mark.parametrize("M,R,dynamic", [(300, 410, False), (9000, 11000, False), (700, 900, True)])
def test_seam_s_function_matches_the_torch_statement(M, R, dynamic):
    torch.manual_seed(M)                    # (nn.Linear's default init below)
    rows, idx, scale, row_slot, g = _seam_s_case(M, R, seed=M + R, nan_tail=333 if dynamic else 0)
    res = torch.randn(1, M, 256, generator=g)
    w0, b0 = torch.randn(256, 256, generator=g) * 0.06, torch.randn(256, generator=g) * 0.1
    fc1, fc2 = nn.Linear(256, 512).to(DEV), nn.Linear(512, 256).to(DEV)
    n0, n1 = nn.LayerNorm(256).to(DEV), nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        for n in (n0, n1):
            n.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
            n.bias.copy_(0.1 * torch.randn(256, generator=g))
    gy = torch.randn(1, M, 256, generator=g)
    L = [_leaf(t) for t in (rows, w0, b0, res)]
    nrows = torch.tensor([R], dtype=torch.int32, device=DEV) if dynamic else None
    y = train_ops.seam_s(L[0], L[1], L[2], L[3], n0, fc1, fc2, n1, gather=(idx.to(DEV), scale.to(DEV)),
                         row_slot=row_slot.to(DEV), nrows=nrows)
    assert y is not None
    y.backward(gy.to(DEV))
    # float64 statement (camera mean as the gather it is)
    d = [t.double().clone().requires_grad_(True) for t in (rows[:R], w0, b0, res)]
    P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in
         dict(w1=fc1.weight, b1=fc1.bias, w2=fc2.weight, b2=fc2.bias, g0=n0.weight, be0=n0.bias, g1=n1.weight, be1=n1.bias).items()}
    ii = idx.long()
    pad = torch.cat([d[0], d[0].new_zeros(1, 256)], 0)
    a = (pad[ii[:, 0]] + pad[ii[:, 1]]) * scale.double()[:, None]          # (-1 -> the zero row)
    x = _ln(a @ d[1].t() + d[2] + d[3], P["g0"], P["be0"], n0.eps)
    h = torch.relu(x @ P["w1"].t() + P["b1"])
    yw = _ln(x + h @ P["w2"].t() + P["b2"], P["g1"], P["be1"], n1.eps)
    yw.backward(gy.double())
    _check(y, yw, "y", 1e-5, 1e-4)
    # (the ReLU mask of a hidden unit within fp32 round-off of zero may differ from the float64 statement's: a handful of
    # isolated entries among M x 512 — bounded in L2, loosely in max)
    _check(L[0].grad[:R], d[0].grad, "grad rows", 1e-3, 3e-2)
    for name, a_, b_ in zip(("w0", "b0", "res"), L[1:], d[1:]):
        _check(a_.grad, b_.grad, "grad " + name, 1e-3, 3e-2)
    for name, mod_p in (("w1", fc1.weight), ("b1", fc1.bias), ("w2", fc2.weight), ("b2", fc2.bias), ("g0", n0.weight),
                        ("be0", n0.bias), ("g1", n1.weight), ("be1", n1.bias)):
        _check(mod_p.grad, P[name].grad, "grad " + name, 1e-3, 3e-2)


def test_two_source_linear_function_matches_the_torch_statement():
    g = torch.Generator().manual_seed(5)
    Q, N = 2500, 192
    first, query, pos = (torch.randn(1, Q, 256, generator=g) for _ in range(3))
    w, b = torch.randn(N, 512, generator=g) * 0.05, torch.randn(N, generator=g) * 0.1
    gy = torch.randn(1, Q, N, generator=g)
    L = [_leaf(t) for t in (first, query, pos, w, b)]
    y = train_ops.two_source_linear(*L)
    y.backward(gy.to(DEV))
    d = [t.double().clone().requires_grad_(True) for t in (first, query, pos, w, b)]
    yw = torch.cat([d[0], d[1] + d[2]], -1) @ d[3].t() + d[4]
    yw.backward(gy.double())
    _check(y, yw, "y", 1e-5, 1e-4)
    for name, a, b_ in zip(("first", "query", "pos", "w", "b"), L, d):
        _check(a.grad, b_.grad, "grad " + name)
    # a detached history gets no gradient and costs none
    L2 = [_leaf(t) for t in (first, query, pos, w, b)]
    L2[0] = L2[0].detach()
    train_ops.two_source_linear(*L2).backward(gy.to(DEV))
    _check(L2[1].grad, d[1].grad, "grad query (detached history)")


def test_grouped_linear_function_matches_the_torch_statement():
    g = torch.Generator().manual_seed(6)
    Lyr, Q = 3, 3000
    hist, cur = torch.randn(1, Q, 256, generator=g), torch.randn(1, Q, 256, generator=g)
    w, b = torch.randn(Lyr * 256, 256, generator=g) * 0.06, torch.randn(Lyr * 256, generator=g) * 0.1
    gys = [torch.randn(2 * Q, 256, generator=g) for _ in range(Lyr)]
    cur_d, w_d, b_d = _leaf(cur), _leaf(w), _leaf(b)
    ys = train_ops.grouped_linear([hist.to(DEV).reshape(-1, 256), cur_d.reshape(-1, 256)], w_d, b_d, Lyr, "tsa_value_proj")
    torch.autograd.backward(list(ys[:2]), [t.to(DEV) for t in gys[:2]])           # (the third output takes no part in the loss)
    c64, w64, b64 = (t.double().clone().requires_grad_(True) for t in (cur, w, b))
    x = torch.cat([hist.double().reshape(-1, 256), c64.reshape(-1, 256)], 0)
    yw = (x @ w64.t() + b64).view(2 * Q, Lyr, 256).unbind(1)
    torch.autograd.backward(list(yw[:2]), [t.double() for t in gys[:2]])
    for i in range(Lyr):
        _check(ys[i], yw[i], f"y{i}", 1e-5, 1e-4)
    _check(cur_d.grad, c64.grad, "grad current rows")
    _check(w_d.grad, w64.grad, "grad w")
    _check(b_d.grad, b64.grad, "grad b")


def _grads(enc, q, f, kw, gout):
    enc.zero_grad(set_to_none=True)
    qd, fd = q.clone().requires_grad_(True), f.clone().requires_grad_(True)
    out = enc(qd, fd, fd, **kw)
    out.backward(gout)
    return out.detach(), {"bev_query": qd.grad, "feats": fd.grad, **{k: p.grad for k, p in enc.named_parameters()}}


@pytest.mark.parametrize("name,temporal,storage", [("micro4", True, torch.float32), ("micro4", False, torch.float32),
                                                   ("tiny", True, torch.float32), ("micro4", True, torch.bfloat16)])
def test_fast_training_path_equals_the_per_op_path(name, temporal, storage):
    """Same encoder, same frame: the chain-kernel autograd path (device-side plan, row count on the device through
    forward and backward) against the per-op autograd path of rounds 2-3 (``train_chain=False``: host-sized plan, one
    Function per Linear / LayerNorm).  Both were / are checked against the oracle elsewhere; here they must agree to
    GEMM-association round-off on the output and on every gradient."""
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=4, temporal=temporal, device=DEV)
    gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(9)).to(DEV)
    before = train_ops.stats()
    with ops.using(value_storage=storage, gemm="bf16" if storage == torch.bfloat16 else "split"):
        out_f, g_f = _grads(enc, q, f, kw, gout)
        after = train_ops.stats()
        with ops.using(train_chain=False):
            out_s, g_s = _grads(enc, q, f, kw, gout)
    L = len(enc.layers)
    assert after["seam_s"] - before["seam_s"] == L and after["seam_t"] - before["seam_t"] == L, (before, after)
    assert train_ops.stats()["seam_s"] == after["seam_s"], "train_chain=False still took the chain kernels"
    bf = storage == torch.bfloat16
    _check(out_f, out_s, "output", 2e-2 if bf else 2e-5, 5e-2 if bf else 2e-4)
    for k in g_s:
        assert g_f[k] is not None and g_s[k] is not None, k
        # (bilinear sampling is piecewise linear: a tap may flip sides between two forward round-offs — L2, not max)
        e2, _ = _rel(g_f[k], g_s[k])
        assert e2 < (8e-2 if bf else 3e-2), f"grad {k}: relative L2 {e2:.2e}"


@pytest.mark.parametrize("name,storage", [("micro4", torch.float32), ("tiny", torch.float32), ("micro4", torch.bfloat16)])
def test_locations_saved_by_the_forward_kernel_equal_the_recomputed_ones(name, storage):
    """``bevmsda_fused_forward_rows_save_*``: the SCA sampling kernel of the training forward writes the sampling
    locations / attention weights of its rows; the backward that reads them must produce what the backward that
    recomputes them (``bevmsda_frontend_expand_rows_f32``, ``fused_save = False``) produces — same formulas, same
    kernels downstream."""
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=6, temporal=True, device=DEV)
    gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(3)).to(DEV)
    with ops.using(value_storage=storage, gemm="bf16" if storage == torch.bfloat16 else "split"):
        out_a, g_a = _grads(enc, q, f, kw, gout)
        with ops.using(fused_save=False):
            out_b, g_b = _grads(enc, q, f, kw, gout)
    # (two instantiations of the sampling kernel: the same arithmetic, scheduled apart — last-bit differences that the
    # layers carry along: measured 4e-6 relative on the gradients, tools/dbg_save.py)
    bf = storage == torch.bfloat16
    torch.testing.assert_close(out_a, out_b, rtol=0, atol=2e-2 if bf else 2e-5)   # (bf16: a rounding of a value may flip)
    for k in g_b:
        e2, _ = _rel(g_a[k], g_b[k])
        assert e2 < (1e-2 if bf else 1e-4), f"grad {k}: relative L2 {e2:.2e}"


@pytest.mark.parametrize("name,storage", [("micro4", torch.float32), ("small4", torch.float32), ("micro4", torch.bfloat16)])
def test_locations_recomputed_by_the_backward_kernels_equal_the_saved_ones(name, storage):
    """``fused_save = 2`` (round 6, opt-in: less memory, the same time): the training forward keeps its attention weights only and the backward kernels
    (``bevmsda_backward_rows_offs_*``) recompute every sampling location from the projection rows with the forward's own two
    operations — the same bits as the locations ``fused_save = 1`` writes and reads back, so outputs and every gradient are
    equal, not close."""
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=8, temporal=True, device=DEV)
    gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(4)).to(DEV)
    with ops.using(value_storage=storage, gemm="bf16" if storage == torch.bfloat16 else "split"):
        with ops.using(fused_save=2):
            out_a, g_a = _grads(enc, q, f, kw, gout)
        with ops.using(fused_save=1):
            out_b, g_b = _grads(enc, q, f, kw, gout)
    assert torch.equal(out_a, out_b)
    for k in g_b:
        # (grad_value is summed by atomics: run-to-run association differs in the last bits, with either mode; with bf16
        # operands a last bit that flips a rounding downstream shows as 6e-6 on a box of this round's last visit)
        e2, _ = _rel(g_a[k], g_b[k])
        assert e2 < (5e-5 if storage == torch.bfloat16 else 2e-6), f"grad {k}: relative L2 {e2:.2e}"


def test_training_step_replays_from_a_hip_graph():
    """A complete forward + backward of the encoder captured in ONE HIP graph (no host read anywhere: device-side plan,
    row count read by the kernels) and replayed with NEW camera matrices: output and gradients equal the eager step on
    the same matrices."""
    name = "micro4"
    enc, _ = build_pair(name, device=DEV)
    q, f, kw = S.make_inputs(name, seed=7, temporal=True, device=DEV)
    import numpy as np
    base = torch.tensor(np.asarray(kw["img_metas"][0]["lidar2img"]), dtype=torch.float32, device=DEV)
    l2i = base.clone()
    kw = dict(kw, img_metas=[dict(lidar2img=l2i, img_shape=kw["img_metas"][0]["img_shape"])])
    gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(2)).to(DEV)
    qd, fd = q.clone().requires_grad_(True), f.clone().requires_grad_(True)

    def step():
        enc.zero_grad(set_to_none=True)
        qd.grad = fd.grad = None
        out = enc(qd, fd, fd, **kw)
        out.backward(gout)
        return out.detach()

    for _ in range(2):
        step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = step()
    captured = {"q": qd.grad, "f": fd.grad, **{k: p.grad for k, p in enc.named_parameters()}}
    # new camera matrices: a small yaw of the rig changes the visible rows (and their count)
    yaw = 0.03
    T = torch.eye(4, device=DEV)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = np.cos(yaw), -np.sin(yaw), np.sin(yaw), np.cos(yaw)
    l2i.copy_(base @ T)
    graph.replay()
    torch.cuda.synchronize()
    got_out = out_g.clone()
    got = {k: v.clone() for k, v in captured.items()}
    want_out = step().clone()                       # eager, same (new) matrices
    want = {"q": qd.grad, "f": fd.grad, **{k: p.grad for k, p in enc.named_parameters()}}
    _check(got_out, want_out, "output", 1e-6, 1e-5)
    for k in want:
        e2, _ = _rel(got[k], want[k])
        assert e2 < 1e-3, f"grad {k}: graph replay vs eager relative L2 {e2:.2e}"     # (atomics: summation order)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_multi_problem_weight_gradient_matches_float64(variant, mode):
    """``bevmsda_linear_wgrad_multi_f32``: three problems over the same rows in one launch (a ragged row count, N = 192
    with a partial tile, row views with a stride, one problem without bias) against float64; variant 0 = bf16 planes +
    transposing LDS reads (csrc/wgrad_tr.h), 1 = the first kernel's gathered fragments, 2 = 256 x 128 tiles on eight wavefronts with two
    LDS stages (round 6; forced here with a workgroup target, since the library picks it only where its grid fills the chip)."""
    g = torch.Generator().manual_seed(17 + variant)
    M = 4999
    G1, X1 = torch.randn(M, 256, generator=g), torch.randn(M, 512, generator=g)
    G2, X2 = torch.randn(M, 192, generator=g), torch.randn(M, 256, generator=g)
    big = torch.randn(M, 768, generator=g)
    G3, X3 = big[:, 256:512], torch.randn(M, 256, generator=g)            # a column block of a wider matrix (ld = 768)
    probs_cpu = [(G1, X1, True), (G2, X2, True), (G3, X3, False)]
    dev_probs, outs = [], []
    bigd = big.to(DEV)
    for i, (G, X, bias) in enumerate(probs_cpu):
        Gd = bigd[:, 256:512] if i == 2 else G.to(DEV)
        gw = torch.zeros(G.shape[1], X.shape[1], device=DEV)
        gb = torch.zeros(G.shape[1], device=DEV) if bias else None
        dev_probs.append((Gd, X.to(DEV), gw, gb))
        outs.append((gw, gb))
    with ops.using(gemm=mode, wgrad_variant=variant, wgrad_workgroups=224 if variant == 2 else 0):
        train_ops._wgrad_multi(dev_probs, "test_dw")
    torch.cuda.synchronize()
    tol = (2e-5, 1e-4) if mode == "split" else (6e-3, 3e-2)
    for (G, X, bias), (gw, gb) in zip(probs_cpu, outs):
        want = G.double().t() @ X.double()
        _check(gw, want, "grad_w", *tol)
        if bias:
            _check(gb, G.double().sum(0), "grad_b", 1e-5, 1e-4)


def _fixed_masks(monkeypatch, seed):
    """``train_ops.dropout_scale`` replaced by seeded Bernoulli scale tensors (CPU generator: the float64 statement uses
    the same ones); returns the list the draws are appended to."""
    drawn = []
    g = torch.Generator().manual_seed(seed)

    def scale(shape, p, device):
        t = (torch.rand(shape, generator=g) >= p).float() / (1.0 - p)
        drawn.append(t)
        return t.to(device)
    monkeypatch.setattr(train_ops, "dropout_scale", scale)
    return drawn


def test_seam_functions_with_active_dropout(monkeypatch):
    """train() mode: the three nn.Dropout sites of the SCA seam (attention output, FFN hidden, FFN output) and the one of
    the TSA seam as scale tensors inside the chain kernels, forward and backward, against the float64 statements with
    the same masks."""
    drawn = _fixed_masks(monkeypatch, 3)
    torch.manual_seed(5)
    M, R = 1500, 2100
    rows, idx, scale, row_slot, g = _seam_s_case(M, R, seed=77)
    res = torch.randn(1, M, 256, generator=g)
    w0, b0 = torch.randn(256, 256, generator=g) * 0.06, torch.randn(256, generator=g) * 0.1
    fc1, fc2 = nn.Linear(256, 512).to(DEV), nn.Linear(512, 256).to(DEV)
    n0, n1 = nn.LayerNorm(256).to(DEV), nn.LayerNorm(256).to(DEV)
    gy = torch.randn(1, M, 256, generator=g)
    L = [_leaf(t) for t in (rows, w0, b0, res)]
    y = train_ops.seam_s(L[0], L[1], L[2], L[3], n0, fc1, fc2, n1, gather=(idx.to(DEV), scale.to(DEV)),
                         row_slot=row_slot.to(DEV), drop_p=(0.1, 0.2, 0.3))
    assert y is not None and len(drawn) == 3
    y.backward(gy.to(DEV))
    m0, mh, m1 = (t.double().reshape(M, -1) for t in drawn)
    d = [t.double().clone().requires_grad_(True) for t in (rows, w0, b0, res)]
    P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in
         dict(w1=fc1.weight, b1=fc1.bias, w2=fc2.weight, b2=fc2.bias, g0=n0.weight, be0=n0.bias, g1=n1.weight, be1=n1.bias).items()}
    ii = idx.long()
    pad = torch.cat([d[0], d[0].new_zeros(1, 256)], 0)
    a = (pad[ii[:, 0]] + pad[ii[:, 1]]) * scale.double()[:, None]
    x = _ln((a @ d[1].t() + d[2]) * m0 + d[3].reshape(M, 256), P["g0"], P["be0"], n0.eps)
    h = torch.relu(x @ P["w1"].t() + P["b1"]) * mh
    yw = _ln(x + (h @ P["w2"].t() + P["b2"]) * m1, P["g1"], P["be1"], n1.eps)
    yw.backward(gy.double().reshape(M, 256))
    _check(y.reshape(M, 256), yw, "y", 1e-5, 1e-4)
    # (ReLU units within fp32 round-off of zero flip against the float64 statement: isolated entries, amplified by
    # 1 / (1 - p) and, with 1,500 rows, a larger share of every column sum than in the 9,000-row case above: measured
    # 1.3e-3 .. 3.2e-3 relative L2 over runs; a wrong mask or scale would show as O(0.1 .. 1))
    _check(L[0].grad, d[0].grad, "grad rows", 1e-2, 1e-1)
    for name, a_, b_ in zip(("w0", "b0", "res"), L[1:], d[1:]):
        _check(a_.grad, b_.grad, "grad " + name, 1e-2, 1e-1)
    for name, mod_p in (("w1", fc1.weight), ("b1", fc1.bias), ("w2", fc2.weight), ("b2", fc2.bias), ("g0", n0.weight),
                        ("be0", n0.bias), ("g1", n1.weight), ("be1", n1.bias)):
        _check(mod_p.grad, P[name].grad, "grad " + name, 1e-2, 1e-1)

    # the TSA seam
    drawn.clear()
    rows_t = torch.randn(1, M, 256, generator=g)
    w1, b1 = torch.randn(768, 256, generator=g) * 0.06, torch.randn(768, generator=g) * 0.1
    gx, gp = torch.randn(1, M, 256, generator=g), torch.randn(M, 768, generator=g)
    Lt = [_leaf(t) for t in (rows_t, w0, b0, res, w1, b1)]
    n0.zero_grad()
    xo, po = train_ops.seam_t(Lt[0], Lt[1], Lt[2], Lt[3], n0, Lt[4], Lt[5], drop_p=0.25)
    assert len(drawn) == 1
    torch.autograd.backward([xo, po], [gx.to(DEV), gp.to(DEV)])
    mt = drawn[0].double().reshape(M, 256)
    dt = [t.double().clone().requires_grad_(True) for t in (rows_t, w0, b0, res, w1, b1)]
    gam, bet = n0.weight.detach().double().cpu().requires_grad_(True), n0.bias.detach().double().cpu().requires_grad_(True)
    xw = _ln((dt[0].reshape(M, 256) @ dt[1].t() + dt[2]) * mt + dt[3].reshape(M, 256), gam, bet, n0.eps)
    pw = xw @ dt[4].t() + dt[5]
    torch.autograd.backward([xw, pw], [gx.double().reshape(M, 256), gp.double()])
    _check(xo.reshape(M, 256), xw, "x", 1e-5, 1e-4)
    for name, a_, b_ in zip(("rows", "w0", "b0", "res", "w1", "b1"), Lt, dt):
        _check(a_.grad.reshape(b_.grad.shape), b_.grad, "TSA seam grad " + name)
    _check(n0.weight.grad, gam.grad, "TSA seam grad gamma")


def test_train_mode_takes_the_chain_kernels_with_dropout_active():
    """``train()`` with p = 0.1 everywhere (the reference's training configuration): the layer still runs on the chain
    kernels (seam counters move), the output differs from eval mode, gradients are finite.  Parity of this mode against
    the oracle with shared deterministic masks: tests/test_encoder_gpu.py::test_train_mode_with_active_dropout_on_the_gpu."""
    enc, _ = build_pair("micro4", device=DEV)
    q, f, kw = S.make_inputs("micro4", seed=4, temporal=True, device=DEV)
    with torch.no_grad():
        ref = enc(q, f, f, **kw)
    enc.train()
    before = train_ops.stats()
    qd = q.clone().requires_grad_(True)
    torch.manual_seed(0)
    out = enc(qd, f, f, **kw)
    after = train_ops.stats()
    assert after["seam_s"] - before["seam_s"] == len(enc.layers) and after["seam_t"] - before["seam_t"] == len(enc.layers)
    assert (out - ref).abs().max() > 1e-3
    out.sum().backward()
    assert torch.isfinite(qd.grad).all() and all(torch.isfinite(p.grad).all() for p in enc.parameters())


@pytest.mark.parametrize("case", ["bs2", "three_cameras"])
def test_fast_training_path_with_a_batch_and_with_three_cameras_per_query(case):
    """Fast path == per-op path on output and gradients (i) for batch size 2 (the TSA value of [history ; current] stays
    one stacked tensor, the literal bs > 1 quirks of the reference) and (ii) on a rig whose cameras 0, 1, 2 coincide:
    every visible query has three rows — the fold of the third row into the first before the two-row gather, its
    gradient (all rows of a query take the query's gradient) and the walk over the full row table in the backward."""
    name = "micro4"
    enc, _ = build_pair(name, device=DEV)
    bs = 2 if case == "bs2" else 1
    q, f, kw = S.make_inputs(name, seed=6, temporal=True, bs=bs, device=DEV)
    if case == "three_cameras":
        mats = kw["img_metas"][0]["lidar2img"]
        kw["img_metas"][0]["lidar2img"] = [mats[0], mats[0].copy(), mats[0].copy(), mats[3], mats[4], mats[5]]
    gout = torch.randn(bs, q.shape[0], 256, generator=torch.Generator().manual_seed(1)).to(DEV)
    before = train_ops.stats()
    out_f, g_f = _grads(enc, q, f, kw, gout)
    assert train_ops.stats()["seam_s"] - before["seam_s"] == len(enc.layers)
    with ops.using(train_chain=False):
        out_s, g_s = _grads(enc, q, f, kw, gout)
    _check(out_f, out_s, "output", 2e-5, 2e-4)
    for k in g_s:
        e2, _ = _rel(g_f[k], g_s[k])
        assert e2 < 3e-2, f"grad {k}: relative L2 {e2:.2e}"


@pytest.mark.parametrize("which", ["sca", "tsa"])
def test_second_consumer_of_the_hoisted_values_keeps_its_gradient(which):
    """ADVICE r4 / VERDICT r5: an auxiliary loss on a hoisted value tensor is a SECOND consumer of a grouped projection's
    output; its gradient reaches the projection's backward summed with the sampling operator's placeholder and must be
    added to the gradient the operator deposited in the sink — not dropped.  Checked on the value-projection weight
    gradients against the per-op path (``train_chain=False``: plain autograd accumulation) with the same auxiliary loss."""
    enc, _ = build_pair("micro4", device=DEV)
    q, f, kw = S.make_inputs("micro4", seed=4, temporal=True, device=DEV)
    gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(9)).to(DEV) * 1e-2
    taken = {}
    orig = type(enc).hoisted_value_projections_autograd

    def spy(self, value, tsa_value):
        sca, tsa = orig(self, value, tsa_value)
        taken["vals"] = sca if which == "sca" else tsa
        return sca, tsa

    def run(fast):
        enc.zero_grad(set_to_none=True)
        taken.clear()
        out = enc(q, f, f, **kw)
        loss = (out * gout).sum()
        if fast:
            assert taken.get("vals") is not None, "the fast path did not hoist the value projections"
            aux_src = taken["vals"][1]
        else:
            # per-op path: the same tensor = layer 1's own value projection of the same input
            att = enc.layers[1].attentions[1 if which == "sca" else 0]
            vp = (att.deformable_attention if which == "sca" else att).value_proj
            if which == "sca":
                src = f.reshape(-1, 256)
            else:
                src = torch.cat([kw["prev_bev"].reshape(-1, 256), q.reshape(-1, 256)], 0)
            aux_src = torch.nn.functional.linear(src, vp.weight, vp.bias)
        w_aux = torch.linspace(-1, 1, aux_src.numel(), device=DEV).view(aux_src.shape)
        (loss + (aux_src.float() * w_aux).sum() * 1e-3).backward()
        att = enc.layers[1].attentions[1 if which == "sca" else 0]
        vp = (att.deformable_attention if which == "sca" else att).value_proj
        return vp.weight.grad.clone(), vp.bias.grad.clone()

    import unittest.mock as mock
    with mock.patch.object(type(enc), "hoisted_value_projections_autograd", spy):
        gw_f, gb_f = run(True)
    with ops.using(train_chain=False):
        gw_s, gb_s = run(False)
    # the auxiliary term dominates these two gradients: dropping it (the round-5 behaviour) is a relative error of ~1
    e_w, _ = _rel(gw_f, gw_s)
    e_b, _ = _rel(gb_f, gb_s)
    assert e_w < 2e-2 and e_b < 2e-2, (e_w, e_b)


def test_training_step_rebuilds_its_weight_images_with_one_launch():
    """Round 6: from the second differentiable step on, the weight images of the trainable weights are
    rebuilt from the weights' current values by ONE launch at the start of the step (``ops.begin_training_step``) instead of one
    launch per image — and an in-place weight update between steps (an optimizer step) is seen: same gradients as with the
    batching switched off, eagerly and from a captured graph of the step."""
    name = "micro4"
    enc, _ = build_pair(name, device=DEV)
    for p in enc.parameters():
        p.requires_grad_(True)
    q, f, kw = S.make_inputs(name, seed=7, temporal=True, device=DEV)
    gout = torch.randn(1, q.shape[0], 256, generator=torch.Generator().manual_seed(2)).to(DEV)

    def step():
        enc.zero_grad(set_to_none=True)
        out = enc(q, f, f, **kw)
        out.backward(gout)
        return out.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters()}

    def sgd(scale):
        with torch.no_grad():
            for i, p in enumerate(enc.parameters()):
                p.add_(torch.full_like(p, scale * (1 + i % 3)))

    ops.set_training_image_batching(True)
    try:
        s0 = ops.training_image_stats()
        step()                                           # registers the images as they are packed
        s1 = ops.training_image_stats()
        assert s1["single_launches"] - s0["single_launches"] >= 10 and s1["images"] >= 10, (s0, s1)
        step()
        s2 = ops.training_image_stats()
        assert s2["multi_launches"] == s1["multi_launches"] + 1
        left = s2["single_launches"] - s1["single_launches"]
        assert left <= 0.4 * (s1["single_launches"] - s0["single_launches"]), \
            f"{left} images still packed one by one in the second step (first step: {s1['single_launches'] - s0['single_launches']}): {s2['unregistered']}"
        print("single-image launches: first step", s1["single_launches"] - s0["single_launches"], "second step", left, s2["unregistered"])
        sgd(1e-3)                                        # an optimizer step between two training steps
        out_b, g_b = step()
        ops.set_training_image_batching(False)
        out_s, g_s = step()
        _check(out_b, out_s, "output after a weight update", 1e-6, 1e-5)
        for k in g_s:
            e2, _ = _rel(g_b[k], g_s[k])
            assert e2 < 1e-3, f"grad {k}: batched vs single image rebuild, relative L2 {e2:.2e}"
        # ... and inside a captured graph: capture, update the weights in place, replay — with the one-launch rebuild and
        # without it (image by image).  Until round 6 a captured training step FROZE the images of the weights that reach the
        # kernels as the parameter objects themselves (ops._cache_ok: the autograd Functions run with grad mode off):
        # 4e-3 on the output, 7 % on gradients after one update (tools/probes/graph_update_check.py).
        for batching in (True, False):
            ops.set_training_image_batching(batching)
            step()
            step()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            holder = {}

            def captured_step():
                enc.zero_grad(set_to_none=True)
                out = enc(q, f, f, **kw)
                out.backward(gout)
                holder["out"] = out.detach()
                holder["g"] = {k: p.grad for k, p in enc.named_parameters()}
            before = ops.training_image_stats()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                captured_step()
            after = ops.training_image_stats()
            assert after["multi_launches"] == before["multi_launches"] + int(batching), "the one-launch rebuild was not captured"
            sgd(-2e-3 if batching else 1.5e-3)
            graph.replay()
            torch.cuda.synchronize()
            got_out, got = holder["out"].clone(), {k: v.clone() for k, v in holder["g"].items()}
            ops.set_training_image_batching(False)
            want_out, want = step()
            _check(got_out, want_out, f"graph replay after a weight update (batching {batching})", 1e-6, 1e-5)
            for k in want:
                e2, _ = _rel(got[k], want[k])
                assert e2 < 1e-3, f"grad {k}: graph replay after a weight update vs eager (batching {batching}), relative L2 {e2:.2e}"
            del graph
    finally:
        ops.set_training_image_batching(True)
