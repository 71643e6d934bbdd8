"""The build recipe of libbevmsda.so (bevformer_amd/build.py), checked on the BUILT objects (no GPU).

The sampling-backward translation unit (csrc/bevmsda_capi_backward.hip) must contain no packed fp32 arithmetic:
with it ``msda_gradloc_d32_kernel`` computed, about once in fifty training passes on a GPU shared with a second process,
one wrong grad_loc_y from bit-identical inputs (profiles/r5/r5_ddp_forensics.txt).  The flag that guarantees it is
``-fno-slp-vectorize`` for that unit only; this test disassembles the unit's gfx950 code object and looks."""
import os
import re
import shutil
import subprocess

import pytest

from bevformer_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"
PACKED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")


def _disassemble(src, tmp_path):
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    build.build_library()
    obj = shutil.copy(build._obj(src), tmp_path / "unit.o")
    subprocess.run([objdump, "--offloading", str(obj)], check=True, capture_output=True, cwd=tmp_path)
    co = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert len(co) == 1, co
    assert "gfx950" in co[0]
    return subprocess.run([objdump, "-d", str(tmp_path / co[0])], check=True, capture_output=True, text=True).stdout


def test_sampling_backward_unit_has_no_packed_fp32_math(tmp_path):
    assert build.EXTRA_FLAGS.get("bevmsda_capi_backward.hip") == ["-fno-slp-vectorize"]
    text = _disassemble("bevmsda_capi_backward.hip", tmp_path)
    kernels = set(re.findall(r"<(_ZN7bevmsda\w+)>:", text))
    assert any("msda_gradloc_d32_kernel" in k for k in kernels) and any("msda_gradvalue_sort_kernel" in k for k in kernels)
    assert not any("msda_fused" in k or "msda_fwd" in k for k in kernels), "forward kernels belong to the other unit"
    hits = PACKED.findall(text)
    assert not hits, f"{len(hits)} packed fp32 instructions in the sampling-backward unit"


def test_forward_unit_keeps_packed_fp32_math(tmp_path):
    """... and the forward unit keeps them: the fused SCA sampling kernel is 8 % slower without (profiles/r5/r5q_slp_ab.txt)."""
    text = _disassemble("bevmsda_capi.hip", tmp_path)
    kernels = set(re.findall(r"<(_ZN7bevmsda\w+)>:", text))
    assert any("msda_fused_d32" in k for k in kernels)
    assert not any("gradloc" in k or "msda_bwd" in k for k in kernels), "backward kernels belong to the other unit"
    assert len(PACKED.findall(text)) > 1000
