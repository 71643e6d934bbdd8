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
