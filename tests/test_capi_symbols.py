"""The C-ABI library loads on a machine without a GPU and exports every entry
point include/bevmsda.h declares, with the ABI version the host expects.  No
compute call is made here (that is the `-m gpu` suite)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bevmsda.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bevmsda_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from bevformer_amd import build
    if build.is_stale():
        build.build_library()
    return build.LIB_PATH


def test_header_declares_the_operator_pair():
    names = _declared()
    for want in ("bevmsda_forward_f32", "bevmsda_backward_f32", "bevmsda_forward_bf16",
                 "bevmsda_backward_bf16", "bevmsda_abi_version", "bevmsda_error_string"):
        assert want in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/bevmsda.h but not exported"
    from bevformer_amd import _lib
    assert lib.bevmsda_abi_version() == _lib.ABI_VERSION
    # host binding table and header agree (nothing bound that is not declared, and vice versa)
    assert sorted(_lib.SIGNATURES) == [n for n in _declared()]


def test_exports_are_plain_c_names(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True,
                         check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for name in _declared():
        assert name in exported


def test_error_strings_and_host_argument_checks_without_gpu(lib_path):
    lib = ctypes.CDLL(lib_path)
    lib.bevmsda_error_string.restype = ctypes.c_char_p
    assert lib.bevmsda_error_string(0) == b"ok"
    for code in range(-1, -10, -1):
        assert lib.bevmsda_error_string(code)  # never NULL
    # argument validation happens before any device work: negative dims are
    # rejected and an all-empty problem is a no-op, with no GPU present
    from bevformer_amd import _lib
    h = _lib.load(lib_path)
    assert h.bevmsda_forward_f32(None, None, None, None, None, -1, 0, 8, 32, 1, 0, 4, None, None) \
        == -2
    assert h.bevmsda_forward_f32(None, None, None, None, None, 0, 0, 8, 32, 1, 0, 4, None, None) \
        == 0


def test_projection_and_prologue_argument_checks_without_gpu(lib_path):
    """The entry points added for the projections and the encoder's caller validate their
    arguments before any device work (error codes of include/bevmsda.h), so a bad call is a
    return value, never a launch.  Pointers are fake but non-NULL where a check must get past
    the NULL test; no kernel runs in this test."""
    from bevformer_amd import _lib
    h = _lib.load(lib_path)
    OK, NULLP, SHAPE, MISAL, OPT, UNSUP = 0, -1, -2, -4, -6, -7
    fake = ctypes.c_void_p(0x1000)                # 16-byte aligned, never dereferenced
    odd = ctypes.c_void_p(0x1004)

    def desc(**kw):
        base = dict(M=8, ldx0=64, ldw=64, ldy=128, N=128, K0=64, K1=0)
        base.update(kw)
        return ctypes.byref(_lib.LinearDesc(**base))

    lin = h.bevmsda_linear_f32
    assert lin(fake, None, None, None, fake, None, None, fake, None) == NULLP          # no descriptor
    assert lin(fake, None, None, None, None, None, desc(), fake, None) == NULLP        # no weight
    assert lin(fake, None, None, None, fake, None, desc(M=-1), fake, None) == SHAPE
    assert lin(fake, None, None, None, fake, None, desc(M=0), fake, None) == OK        # empty: no-op
    assert lin(fake, None, None, None, fake, None, desc(K0=48), fake, None) == UNSUP   # K % 32
    assert lin(fake, None, None, None, fake, None, desc(ldx0=66), fake, None) == UNSUP
    assert lin(fake, None, None, None, fake, None, desc(ldx0=32), fake, None) == SHAPE  # ld < K
    assert lin(odd, None, None, None, fake, None, desc(), fake, None) == MISAL
    assert lin(fake, None, None, None, fake, None, desc(precision=7), fake, None) == OPT
    assert lin(fake, None, None, None, fake, None, desc(K1=32), fake, None) == NULLP   # x1 missing
    assert lin(fake, None, None, None, fake, None, desc(group_cols=96), fake, None) == UNSUP
    assert lin(fake, None, None, None, fake, None, desc(variant=1 + 12), fake, None) == OPT   # packed variant, fp32 weight
    assert h.bevmsda_linear_packed_f32(fake, None, None, None, None, None, desc(), fake, None) == NULLP
    assert h.bevmsda_linear_packed_f32(fake, None, None, None, fake, None, desc(variant=1 + 2), fake, None) == OPT
    assert h.bevmsda_linear_gather_packed_f32(fake, 64, None, fake, fake, None, desc(), fake, None) == NULLP
    assert h.bevmsda_linear_gather_packed_f32(fake, 64, fake, fake, fake, None, desc(K1=32), fake, None) == SHAPE

    assert h.bevmsda_linear_packed_bytes(256, 256) == 2 * 8 * 2 * 128 * 40 * 2
    assert h.bevmsda_linear_packed_bytes(130, 64) == 2 * 2 * 2 * 128 * 40 * 2          # N padded to 256
    assert h.bevmsda_linear_packed_bytes(256, 48) == 0
    assert h.bevmsda_linear_pack_weight_f32(fake, 64, 0, 64, fake, None) == SHAPE
    assert h.bevmsda_linear_pack_weight_f32(fake, 64, 8, 48, fake, None) == UNSUP
    assert h.bevmsda_linear_pack_weight_f32(None, 64, 8, 64, fake, None) == NULLP

    theta = (ctypes.c_float * 6)(1, 0, 0, 0, 1, 0)
    rot = h.bevmsda_rotate_bev_f32
    assert rot(fake, 256, fake, 256, 4, 4, 256, theta, None) == OPT                     # in place
    assert rot(fake, 256, ctypes.c_void_p(0x2000), 256, 4, 4, 100, theta, None) == UNSUP
    assert rot(fake, 128, ctypes.c_void_p(0x2000), 256, 4, 4, 256, theta, None) == SHAPE
    assert rot(fake, 256, ctypes.c_void_p(0x2000), 256, 0, 4, 256, theta, None) == OK
    assert rot(None, 256, ctypes.c_void_p(0x2000), 256, 4, 4, 256, theta, None) == NULLP
    fl = h.bevmsda_flatten_feats_f32
    assert fl(fake, None, None, fake, 1, 6, 256, 10, 5, 0, None) == SHAPE               # s0 + hw > S
    assert fl(fake, None, None, fake, 1, 6, 100, 10, 10, 0, None) == UNSUP
    assert fl(fake, None, None, fake, 0, 6, 256, 10, 10, 0, None) == OK
    assert fl(None, None, None, fake, 1, 6, 256, 10, 10, 0, None) == NULLP
