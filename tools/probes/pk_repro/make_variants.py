"""Code objects for the packed-fp32 forensics (profiles/r6/r6_pk_forensics.txt).

The sampling-backward translation unit is compiled to gfx950 ASSEMBLY twice — with the SLP vectorizer (the build that
produced the sporadic wrong grad_loc_y of round 5) and without — and the assembly of the SLP build is edited INSIDE
``msda_gradloc_d32_kernel`` only, one hypothesis per variant, then assembled and linked to a stand-alone ``.hsaco`` that
``hunt.py`` loads with hipModuleLoad:

  slp          the packed build, untouched
  noslp        -fno-slp-vectorize (no v_pk_{add,mul,fma}_f32)
  nop_all      s_nop 7 before and after EVERY packed fp32 instruction          (any wait-state hazard around them)
  nop_add      ... only around v_pk_add_f32 (the op_sel source-selecting ones)
  nop_mulfma   ... only around v_pk_mul_f32 / v_pk_fma_f32
  wait_all     s_waitcnt vmcnt(0) lgkmcnt(0) before every packed instruction   (an operand still in flight)

    python tools/probes/pk_repro/make_variants.py [outdir]      (CPU only: hipcc cross-compiles)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "bevformer_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-pass-failed", "--cuda-device-only", "-S"]
KERNEL = re.compile(r"^_ZN7bevmsda23msda_gradloc_d32_kernelI\w+:")
PACKED = re.compile(r"^\s*v_pk_(add|mul|fma)_f32\b")


def device_asm(out, extra=()):
    subprocess.run(["hipcc"] + FLAGS + list(extra) + [os.path.join(CSRC, "bevmsda_capi_backward.hip"), "-o", out],
                   check=True, cwd=CSRC, stderr=subprocess.DEVNULL)


def edit(lines, rule):
    """``rule(kind) -> (before, after)`` lists of instructions for a packed instruction of ``kind`` inside the kernels."""
    out, inside, n = [], False, 0
    for ln in lines:
        if KERNEL.match(ln):
            inside = True
        elif inside and ln.startswith(".Lfunc_end"):
            inside = False
        m = PACKED.match(ln) if inside else None
        if m:
            before, after = rule(m.group(1))
            out += ["\t" + b + "\n" for b in before]
            out.append(ln)
            out += ["\t" + a + "\n" for a in after]
            n += bool(before or after)
        else:
            out.append(ln)
    return out, n


def assemble(asm, hsaco):
    obj = hsaco[:-6] + ".o"
    subprocess.run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950",
                    "-c", asm, "-o", obj], check=True)
    subprocess.run([os.path.join(LLVM, "ld.lld"), "-shared", obj, "-o", hsaco], check=True)
    os.remove(obj)


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    slp_s, noslp_s = os.path.join(outdir, "slp.s"), os.path.join(outdir, "noslp.s")
    if not (os.path.exists(slp_s) and os.path.exists(noslp_s)) or "--fresh" in sys.argv:
        device_asm(slp_s)
        device_asm(noslp_s, ["-fno-slp-vectorize"])
    lines = open(slp_s).readlines()
    nop = ["s_nop 7"]
    rules = {
        "nop_all": lambda k: (nop, nop),
        "nop_add": lambda k: (nop, nop) if k == "add" else ([], []),
        "nop_mulfma": lambda k: (nop, nop) if k != "add" else ([], []),
        "wait_all": lambda k: (["s_waitcnt vmcnt(0) lgkmcnt(0)"], []),
    }
    assemble(slp_s, os.path.join(outdir, "slp.hsaco"))
    assemble(noslp_s, os.path.join(outdir, "noslp.hsaco"))
    for name, rule in rules.items():
        ed, n = edit(lines, rule)
        p = os.path.join(outdir, name + ".s")
        open(p, "w").writelines(ed)
        assemble(p, os.path.join(outdir, name + ".hsaco"))
        os.remove(p)
        print(f"{name}: {n} packed instructions edited")
    packed = sum(1 for ln in edit(lines, lambda k: (["x"], []))[0] if ln == "\tx\n")
    print(f"slp: {packed} packed fp32 instructions inside msda_gradloc_d32_kernel<*>")
    assert not any(PACKED.match(ln) for ln in open(noslp_s)), "the no-SLP build must have no packed fp32 math"


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "tools", "probes", "pk_repro", "hsaco"))
