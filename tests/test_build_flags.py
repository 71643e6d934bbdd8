"""The ISA of libbevmsda.so, checked on the BUILT objects (no GPU).

No code object of the library may contain packed fp32 arithmetic with an ``op_sel`` bit set — a LOW result lane reading the
HIGH half of a source register pair, e.g. ``v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,0]``.  On gfx950 that form
delivered, about once in 150 training passes with a second process busy on the same GPU, a wrong result for one 16-lane
pass (one wrong grad_loc_y from bit-identical inputs; root-caused in round 6 by class-by-class bisection on edited
assembly: profiles/r6/r6_pk_forensics.txt, tools/probes/pk_repro/).  The guarantee comes from the SOURCE
(csrc/scalar_ops.h: the expressions the SLP vectoriser would pair that way are single opaque VALU instructions), not
from a compiler flag, so a maintainer's plain ``hipcc`` build keeps it — and this test is what notices a compiler
upgrade that starts forming the pattern somewhere else.  Packed fp32 math WITHOUT it stays everywhere (the fused SCA sampling kernel is
8 % slower without packed FMAs: profiles/r5/r5q_slp_ab.txt)."""
import os
import re
import shutil
import subprocess

import pytest

from bevformer_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"
PACKED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
# op_sel:[a,b(,c)] with any 1: a low result lane takes the high half of that source (op_sel_hi:[..0..], a high lane taking
# the low half — the broadcast form the sampling kernels live on — is NOT the pattern: scalarising those changed nothing)
CROSS_LOW_FROM_HIGH = re.compile(r"\bv_pk_(?:add|mul|fma)_f32\b.*\bop_sel:\[[01,]*1[01,]*\]")


def _disassemble(src, tmp_path):
    objdump = os.path.join(LLVM, "llvm-objdump")
    # (no skip: a box that cannot disassemble cannot vouch for the build)
    assert os.path.exists(objdump), "llvm-objdump is needed to check the library's ISA"
    build.build_library()
    d = tmp_path / src
    d.mkdir()
    obj = shutil.copy(build._obj(src), d / "unit.o")
    subprocess.run([objdump, "--offloading", str(obj)], check=True, capture_output=True, cwd=d)
    co = [f for f in os.listdir(d) if "amdgcn" in f]
    assert len(co) == 1, co
    assert "gfx950" in co[0]
    return subprocess.run([objdump, "-d", str(d / co[0])], check=True, capture_output=True, text=True).stdout


def test_no_flag_is_involved():
    assert build.EXTRA_FLAGS == {}, "the guarantee must come from the source, not from per-unit compiler flags"
    assert not any("slp" in f for f in build.FLAGS)


@pytest.mark.parametrize("src", build.sources())
def test_no_packed_fp32_instruction_reads_a_high_half_into_a_low_lane(src, tmp_path):
    text = _disassemble(src, tmp_path)
    hits = [ln.strip().split("//")[0] for ln in text.splitlines() if CROSS_LOW_FROM_HIGH.search(ln)]
    assert not hits, f"{len(hits)} packed fp32 instructions with an op_sel bit set in {src}, e.g. {hits[:3]}"


def test_every_unit_was_looked_at_and_packed_math_is_still_there(tmp_path):
    """The five translation units, and the packed FMAs the forward sampling kernels and the GEMM epilogues rely on."""
    assert set(build.sources()) == {"bevmsda_capi.hip", "bevmsda_capi_backward.hip", "bevmsda_frontend.hip",
                                    "bevmsda_linear.hip", "bevmsda_plan.hip"}
    fwd = _disassemble("bevmsda_capi.hip", tmp_path)
    kernels = set(re.findall(r"<(_ZN7bevmsda\w+)>:", fwd))
    assert any("msda_fused_d32" in k for k in kernels)
    assert not any("gradloc" in k or "msda_bwd" in k for k in kernels), "backward kernels belong to the other unit"
    assert len(PACKED.findall(fwd)) > 1000
    bwd = _disassemble("bevmsda_capi_backward.hip", tmp_path)
    kernels = set(re.findall(r"<(_ZN7bevmsda\w+)>:", bwd))
    assert any("msda_gradloc_d32_kernel" in k for k in kernels) and any("msda_gradvalue_sort_kernel" in k for k in kernels)
    assert not any("msda_fused" in k or "msda_fwd" in k for k in kernels), "forward kernels belong to the other unit"
    assert len(PACKED.findall(bwd)) > 1000, "the backward unit is built with the vectoriser again"


def test_the_pattern_is_recognised():
    bad = "\tv_pk_add_f32 v[66:67], v[68:69], v[66:67] op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]"
    ok = ["\tv_pk_add_f32 v[92:93], v[66:67], v[28:29] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
          "\tv_pk_fma_f32 v[8:9], v[50:51], v[50:51], v[4:5] op_sel_hi:[1,1,0]",
          "\tv_pk_mov_b32 v[0:1], v[0:1], v[0:1] op_sel:[1,0]", "\tv_pk_mul_f32 v[2:3], v[2:3], v[2:3]"]
    assert CROSS_LOW_FROM_HIGH.search(bad)
    assert CROSS_LOW_FROM_HIGH.search("\tv_pk_fma_f32 v[28:29], v[28:29], v[22:23], -0.5 op_sel:[1,0,0] op_sel_hi:[0,1,0]")
    assert not any(CROSS_LOW_FROM_HIGH.search(x) for x in ok)
