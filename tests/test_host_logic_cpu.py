"""Host-side helpers of the product package that need no GPU: cached merged / packed parameters,
the rotation matrix handed to the rotate kernel, camera runs of the frame plan, the row order
policy, the row-view helper of the projection op."""
import torch
import torch.nn as nn

from bevformer_amd import ops
from bevformer_amd import synthetic as S
from bevformer_amd.modules import geometry
from oracle import bevformer_cpu as O

from helpers import build_pair


def test_merged_linear_params_cache_follows_parameter_versions():
    owner, a, b = nn.Module(), nn.Linear(8, 4), nn.Linear(8, 6)
    with torch.no_grad():
        w1, b1 = ops.merged_linear_params(owner, a, b)
        w2, _ = ops.merged_linear_params(owner, a, b)
        assert w2 is w1 and w1.shape == (10, 8) and b1.shape == (10,)
        a.weight.mul_(2.0)                                   # in-place update bumps the version
        w3, _ = ops.merged_linear_params(owner, a, b)
        assert w3 is not w1 and torch.equal(w3[:4], a.weight)
    # under autograd the concatenation must stay in the graph: never cached
    w4, _ = ops.merged_linear_params(owner, a, b)
    assert w4.requires_grad and w4 is not w3


def test_rotation_theta_reproduces_the_restated_torchvision_grid():
    """ops.rotation_theta feeds bevmsda_rotate_bev_f32; evaluating the kernel's arithmetic with it
    in torch must give the oracle's source-index map (up to fp32 rounding ties)."""
    for h, w, center, angle in ((12, 10, (5, 6), 4.0), (200, 200, (100, 100), -3.7), (37, 53, (20, 11), 33.0)):
        t = torch.tensor(ops.rotation_theta(angle, center, h, w)).reshape(2, 3)
        bx = torch.arange(w, dtype=torch.float32) + (0.5 - 0.5 * w)
        by = torch.arange(h, dtype=torch.float32) + (0.5 - 0.5 * h)
        gx = (bx[None, :] * t[0, 0] + by[:, None] * t[0, 1]) + t[0, 2]
        gy = (bx[None, :] * t[1, 0] + by[:, None] * t[1, 1]) + t[1, 2]
        ix, iy = torch.round(((gx + 1) * w - 1) / 2), torch.round(((gy + 1) * h - 1) / 2)
        ok = (ix >= 0) & (ix <= w - 1) & (iy >= 0) & (iy <= h - 1)
        mine = torch.where(ok, (iy * w + ix).long(), torch.full((h, w), -1, dtype=torch.long)).flatten()
        want = O.rotate_source_index(h, w, angle, list(center))
        assert (mine != want).float().mean().item() <= 2e-4


def test_camera_runs_and_row_order_policy():
    rb = torch.tensor([0, 0, 0, 2, 2, 5], dtype=torch.int32)
    start, longest = geometry.camera_runs(rb, 6)
    assert start.tolist() == [0, 3, 3, 5, 5, 5, 6] and longest == 3
    start, longest = geometry.camera_runs(rb[:0], 6)
    assert start.tolist() == [0] * 7 and longest == 0
    enc, _ = build_pair("micro")
    assert enc.sca_row_order == "auto"
    with torch.no_grad():
        assert enc.row_order() == "image"
    assert enc.row_order() == "image"          # under autograd too (the LDS-sort backward likes coherent rows)
    enc.sca_row_order = "raster"
    assert enc.row_order() == "raster"
    enc.sca_row_order = "image"
    q, f, kw = S.make_inputs("micro", seed=0)
    with torch.no_grad():
        plan = enc.frame_plan(kw["bev_h"], kw["bev_w"], 1, kw["img_metas"], torch.device("cpu"), torch.float32)
    assert plan.cam_start.tolist()[0] == 0 and plan.cam_start.tolist()[-1] == plan.row_batch.numel()
    assert plan.max_cam_rows == max(plan.hits)
    assert (plan.row_batch[1:] >= plan.row_batch[:-1]).all()          # rows grouped by camera


def test_rows2d_views_without_copies():
    x = torch.randn(2, 5, 64)
    v, ld = ops._rows2d(x, 64)
    assert v.shape == (10, 64) and ld == 64 and v.data_ptr() == x.data_ptr()
    big = torch.randn(7, 96)
    v, ld = ops._rows2d(big[:, 32:], 64)                              # strided rows: still a view
    assert ld == 96 and v.data_ptr() == big[:, 32:].data_ptr()
    v, ld = ops._rows2d(big[:, 1:65], 64)                             # misaligned offset: copied
    assert ld == 64 and v.data_ptr() != big[:, 1:65].data_ptr()


def test_projection_op_declines_on_cpu_and_in_native_mode():
    x, w = torch.randn(4, 64), torch.randn(8, 64)
    saved = ops.gemm_mode()
    try:
        ops.set_gemm_mode("split")
        with torch.no_grad():
            assert ops.linear(x, w) is None                           # CPU tensors: caller uses torch
            torch.testing.assert_close(ops.linear_or_torch(x, w, relu=True), torch.relu(x @ w.t()))
            assert ops.linear_gather_mean(x, torch.zeros(4, 2, dtype=torch.int32), torch.ones(4), w) is None
    finally:
        ops.set_gemm_mode(saved)


def test_modes_are_thread_local_overrides_of_process_defaults():
    """bevformer_amd/modes.py: ``using`` changes what the CALLING thread sees, other threads keep the process
    defaults, nesting restores, and a snapshot re-activated elsewhere (what the autograd Functions do for their
    backward, which the engine runs on its own thread) carries the caller's values."""
    import threading
    import torch
    from bevformer_amd import modes, ops
    base = modes.current().gemm
    seen = {}
    with ops.using(gemm="bf16", value_storage=torch.bfloat16) as m:
        assert modes.current().gemm == "bf16" and ops.value_storage() == torch.bfloat16
        snap = m.snapshot()
        t = threading.Thread(target=lambda: seen.update(other=modes.current().gemm))
        t.start()
        t.join()
        with ops.using(gemm="native"):
            assert modes.current().gemm == "native" and ops.value_storage() == torch.bfloat16
        assert modes.current().gemm == "bf16"

        def engine_thread():
            with modes.activate(snap):
                seen["engine"] = (modes.current().gemm, modes.current().value_storage)
        t = threading.Thread(target=engine_thread)
        t.start()
        t.join()
    assert seen["other"] == base and seen["engine"] == ("bf16", torch.bfloat16)
    assert modes.current().gemm == base and modes.current() is modes.process_defaults()
    import pytest
    with pytest.raises(AttributeError):
        with ops.using(no_such_mode=1):
            pass


def test_positional_encoding_copy_is_cached_only_for_unmodified_memory():
    """``BEVFormerEncoder._contiguous_pos``: the contiguous copy of a transposed positional encoding is reused while the
    caller hands in the same, unmodified memory — and never after an in-place write or for another tensor."""
    import torch
    import bevformer_amd
    from bevformer_amd import synthetic as S
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg("micro")).eval()
    base = torch.randn(1, 8, 12)                       # (bs, C, Q): the layout get_bev_features flattens from
    with torch.no_grad():
        a = enc._contiguous_pos(base.permute(0, 2, 1))
        b = enc._contiguous_pos(base.permute(0, 2, 1))
        assert a is b and a.is_contiguous() and torch.equal(a, base.permute(0, 2, 1))
        base.mul_(2.0)                                 # in-place write: version counter moves
        c = enc._contiguous_pos(base.permute(0, 2, 1))
        assert c is not a and torch.equal(c, base.permute(0, 2, 1))
        other = torch.randn(1, 8, 12)
        d = enc._contiguous_pos(other.permute(0, 2, 1))
        assert torch.equal(d, other.permute(0, 2, 1))
        cont = torch.randn(1, 12, 8)
        assert enc._contiguous_pos(cont) is cont


def test_encoder_runs_under_inference_mode():
    """``torch.inference_mode()``: inference tensors track no version counter, so every cache keyed on ``_version``
    (the contiguous positional encoding, TemporalSelfAttention's reference rows, the merged / packed weights) must
    key on something else there (``_contiguous_pos`` raised on the transposed ``bev_pos`` view)."""
    from helpers import oracle_ops
    enc, sd = build_pair("micro")
    q, f, kw = S.make_inputs("micro", seed=0, temporal=True)
    with oracle_ops(), torch.no_grad():
        want = enc(q, f, f, **kw)
    with oracle_ops(), torch.inference_mode():
        q2, f2, kw2 = S.make_inputs("micro", seed=0, temporal=True)         # inference tensors
        kw2["bev_pos"] = kw2["bev_pos"].permute(1, 2, 0).contiguous().permute(2, 0, 1)   # a transposed view, as through get_bev_features
        assert kw2["bev_pos"].is_inference() and not kw2["bev_pos"].is_contiguous()
        got = enc(q2, f2, f2, **kw2)
        again = enc(q2, f2, f2, **kw2)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(got, again)
    w = torch.nn.Linear(4, 4)
    with torch.inference_mode():
        t = w.weight * 1.0
        assert ops._ver(t) == 0 and ops._ver(w.weight) == w.weight._version


def test_fused_sampling_rejects_unknown_options():
    """``msda_fused`` swallowed every unknown keyword: only the options of the retired LDS-staged kernels
    are still accepted (and ignored); a misspelt one raises before anything is launched."""
    import pytest
    v = torch.zeros(1, 4, 8, 32)
    common = dict(M=8, L=1, P=4, K=1, off_head=8, off_k=0, lg_head=4, lg_k=0, ref_mode=0, vmul=1, vadd=0)
    with pytest.raises(TypeError, match="launch_row"):
        ops.msda_fused(v, None, None, torch.zeros(2, 96), 64, torch.zeros(2, 1, 4, 2), None, launch_row=3, **common)
    with pytest.raises(RuntimeError, match="no CPU path"):      # retired names pass the keyword check (and then: CPU tensor)
        ops.msda_fused(v, None, None, torch.zeros(2, 96), 64, torch.zeros(2, 1, 4, 2), None, cam_start=None, **common)


def test_flattened_linear_params_merge_as_views_with_the_same_gradients():
    """``ops.flatten_linear_params``: the parameters of Linear layers that share their input are re-seated back to back
    (values, Parameter objects and state_dict untouched); ``merged_linear_params`` under autograd is then a view whose
    backward hands every parameter its block — same gradients as the ``cat`` form, and the view follows optimizer steps."""
    import torch
    from bevformer_amd import ops
    torch.manual_seed(0)
    a, b = torch.nn.Linear(8, 6), torch.nn.Linear(8, 4)
    ref_a, ref_b = torch.nn.Linear(8, 6), torch.nn.Linear(8, 4)
    ref_a.load_state_dict(a.state_dict())
    ref_b.load_state_dict(b.state_dict())
    ids = [id(p) for p in list(a.parameters()) + list(b.parameters())]
    assert not ops._adjacent([a.weight, b.weight])
    assert ops.flatten_linear_params(a, b) and ops.flatten_linear_params(a, b)          # idempotent
    assert ids == [id(p) for p in list(a.parameters()) + list(b.parameters())]
    assert all(torch.equal(p, q) for p, q in zip(list(a.parameters()) + list(b.parameters()),
                                                 list(ref_a.parameters()) + list(ref_b.parameters())))
    assert ops._adjacent([a.weight, b.weight]) and ops._adjacent([a.bias, b.bias])

    class Owner:
        pass
    w, bias = ops.merged_linear_params(Owner(), a, b)
    assert w.data_ptr() == a.weight.data_ptr() and bias.data_ptr() == a.bias.data_ptr()
    x = torch.randn(3, 8)
    (x @ w.t() + bias).pow(2).sum().backward()
    (x @ torch.cat([ref_a.weight, ref_b.weight]).t() + torch.cat([ref_a.bias, ref_b.bias])).pow(2).sum().backward()
    for p, q in zip(list(a.parameters()) + list(b.parameters()), list(ref_a.parameters()) + list(ref_b.parameters())):
        torch.testing.assert_close(p.grad, q.grad)
    torch.optim.SGD(list(a.parameters()) + list(b.parameters()), lr=0.1).step()
    w2, _ = ops.merged_linear_params(Owner(), a, b)
    assert torch.equal(w2, torch.cat([a.weight, b.weight])) and not torch.equal(w2, torch.cat([ref_a.weight, ref_b.weight]))
    # a module that was moved / cast afterwards owns separate storages again: the merge is a cat, still correct
    a.double(), b.double()
    a.float(), b.float()
    w3, _ = ops.merged_linear_params(Owner(), a, b)
    assert torch.equal(w3, torch.cat([a.weight, b.weight]))


def test_kernel_selection_table_names_existing_profiles_and_matches_the_library_constants():
    """``ops.KERNEL_SELECTION``: every threshold that picks a projection kernel in one place, each with the profile it was
    measured in (the file must exist) — and the rules that live in the library carry the same numbers there."""
    import os
    import re
    from bevformer_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, (value, what, measured, profile) in ops.KERNEL_SELECTION.items():
        assert os.path.exists(os.path.join(root, profile)), (name, profile)
        assert what and measured
    src = open(os.path.join(root, "bevformer_amd", "csrc", "bevmsda_linear.hip")).read()
    assert int(re.search(r"kLinearPipeMaxRows\s*=\s*(\d+)", src).group(1)) == ops.KERNEL_SELECTION["pipe_max_rows"][0]
    assert int(re.search(r"kChainSmallRows\s*=\s*(\d+)", src).group(1)) == ops.KERNEL_SELECTION["chain_small_rows"][0]
    assert "256LL * 64" in src and ops.KERNEL_SELECTION["chain_mixed_rows"][0] == 256 * 64


def test_clear_weight_caches_drops_every_derived_image():
    """``ops.clear_weight_caches``: the escape hatch for parameters that are inference tensors (no version counter)."""
    import torch
    from bevformer_amd import ops
    lin = torch.nn.Linear(8, 4)
    lin.weight._bevmsda_pack = ("k", torch.zeros(1))
    lin.weight._bevmsda_wt = ("k", torch.zeros(1))
    holder = torch.nn.Module()
    holder.lin = lin
    holder.__dict__["_merged_linear"] = ("k", None, None)
    assert ops.clear_weight_caches(holder) == 3
    assert not hasattr(lin.weight, "_bevmsda_pack") and not hasattr(lin.weight, "_bevmsda_wt")
    assert "_merged_linear" not in holder.__dict__
    assert ops.clear_weight_caches(holder) == 0


def test_ops_package_has_no_dangling_globals_and_keeps_its_substitution_points():
    """``bevformer_amd/ops/`` (round 5: the former ``ops.py``): every global a submodule's code loads exists in that
    submodule (a name left behind by the split would only fail on the path that uses it), every public name sits on the
    package, and a substituted operator (``ops.linear = spy``) is what the OTHER operators call."""
    import builtins
    import dis
    import inspect
    import types
    from bevformer_amd import ops
    from bevformer_amd.ops import _base, chains, gemm, images, prologue, sampling

    def code_objects(co):
        yield co
        for c in co.co_consts:
            if isinstance(c, types.CodeType):
                yield from code_objects(c)

    for mod in (_base, sampling, images, gemm, chains, prologue):
        missing = set()
        for v in vars(mod).values():
            fns = [v] if isinstance(v, types.FunctionType) else \
                [f for f in vars(v).values() if isinstance(f, (types.FunctionType, staticmethod))] if isinstance(v, type) else []
            for f in fns:
                f = f.__func__ if isinstance(f, staticmethod) else f
                if getattr(f, "__module__", None) != mod.__name__:
                    continue
                f = inspect.unwrap(f)               # (once_differentiable / _forward_modes wrappers: the function underneath)
                if f.__globals__ is not vars(mod):
                    continue
                for co in code_objects(f.__code__):
                    for ins in dis.get_instructions(co):
                        if ins.opname == "LOAD_GLOBAL" and ins.argval not in vars(mod) and not hasattr(builtins, ins.argval):
                            missing.add(ins.argval)
        assert not missing, (mod.__name__, sorted(missing))
        for k in vars(mod):
            if not k.startswith("__") and k != "_pkg":
                assert hasattr(ops, k), k
    seen = []
    real = ops.linear
    ops.linear = lambda *a, **k: seen.append("spy") or None
    try:
        import torch
        x, w = torch.zeros(2, 4), torch.zeros(3, 4)
        ops.linear_or_torch(x, w)              # CPU tensors: the spy declines (None), torch takes over
    finally:
        ops.linear = real
    assert seen == ["spy"]


def test_plan_blob_layout_and_views():
    """``geometry.plan_blob_spec`` / ``carve_plan_blob``: the per-frame plan arrays as views of ONE int32 block (so that a
    training step's ``FramePlan.snapshot()`` is one copy): pieces on 256-byte boundaries, disjoint, shaped and typed as asked,
    a clone of the block carved the same way is an independent copy of every piece."""
    import torch
    from bevformer_amd.modules.geometry import carve_plan_blob, plan_blob_spec
    pieces = [("counters", (10,), False), ("row_query", (1000,), False), ("row_ref", (1000, 4, 2), True), ("inv", (2, 50, 1), True)]
    spec, words = plan_blob_spec(pieces)
    assert all(o % 64 == 0 for _, _, _, o in spec) and words % 64 == 0
    blob = torch.zeros(words, dtype=torch.int32)
    v = carve_plan_blob(blob, spec)
    assert v["row_ref"].dtype == torch.float32 and v["row_ref"].shape == (1000, 4, 2) and v["counters"].dtype == torch.int32
    for i, (name, _, _, _) in enumerate(spec):            # disjoint: a write to one piece shows in no other
        v[name].fill_(i + 1)
    for i, (name, _, _, _) in enumerate(spec):
        assert bool((v[name] == i + 1).all()), name
    copy = carve_plan_blob(blob.clone(), spec)
    v["row_ref"].zero_()
    assert bool((copy["row_ref"] == 3).all()) and copy["row_ref"].data_ptr() != v["row_ref"].data_ptr()


def test_captured_weight_images_report_stale_weights(monkeypatch):
    """a graph captured under no_grad freezes the weight images it was captured with; the registry records
    (weight, version, address) at capture so that ``ops.assert_graph_weights_fresh`` can tell a changed weight."""
    import torch
    from bevformer_amd import ops
    from bevformer_amd.ops import images
    ops.release_captured_images()
    w = torch.nn.Parameter(torch.randn(8, 4))
    image = torch.zeros(16, dtype=torch.int16)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    assert images._cached_image(("key", image), w) is image
    assert images._cached_image(("key", image), w) is image        # same image again: one record
    assert len(images._CAPTURED_IMAGES) == 1
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    assert ops.graph_weights_stale() == []
    ops.assert_graph_weights_fresh()
    with torch.no_grad():
        w.add_(1.0)                                                # an optimizer step / load_state_dict between replays
    stale = ops.graph_weights_stale()
    assert len(stale) == 1 and stale[0][0] == (8, 4) and "written" in stale[0][1]
    import pytest
    with pytest.raises(RuntimeError, match="stale"):
        ops.assert_graph_weights_fresh()
    del w
    import gc
    gc.collect()
    assert "freed" in ops.graph_weights_stale()[0][1]
    assert ops.release_captured_images() == 1 and ops.graph_weights_stale() == []
