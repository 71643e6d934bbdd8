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
  gl_noslp     the packed build with ONLY msda_gradloc_d32_* taken from the build without packed math
  sort_noslp   ... with ONLY msda_gradvalue_sort_kernel<*> taken from it
  glsort_noslp ... both
  sc_all / sc_add / sc_addsel / sc_mulfma / sc_sel / sc_nosel
               the packed instructions of ONE class inside the gradloc kernels (all / v_pk_add / v_pk_add with op_sel
               source selection / v_pk_mul + v_pk_fma / any with op_sel / any without) replaced by the two scalar VOP3
               instructions they stand for (bit-identical arithmetic), the rest stay packed   (which class carries it)
  inplace_tmp  every packed instruction whose DESTINATION pair is also a SOURCE pair read with cross-selected halves
               (op_sel / op_sel_hi: the high result reads the low source register or the other way round) reads a COPY of
               that source made by v_mov_b64 into two registers the kernel does not use — the other ~230 packed
               instructions stay as they are                                   (in-place update with swapped halves)

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


INPLACE = re.compile(r"^(\s*v_pk_\w+\s+)v\[(\d+):(\d+)\],\s*(.*)$")


def edit_inplace(lines):
    """The ``inplace_tmp`` variant.  Two spare registers per kernel: next_free_vgpr (and the accum offset) are raised."""
    out, inside, n, top = [], False, 0, None
    for ln in lines:
        if KERNEL.match(ln):
            inside, top = True, None
        elif inside and ln.startswith(".Lfunc_end"):
            inside = False
        m = INPLACE.match(ln) if inside else None
        if m and "op_sel" in m.group(4):
            d0 = int(m.group(2))
            if re.search(r"v\[%d:%d\]" % (d0, d0 + 1), m.group(4)):
                if top is None:
                    top = (TOPS[[k for k in TOPS if k in cur][0]] + 1) // 2 * 2      # (64-bit tuples start at even registers)
                tmp = f"v[{top}:{top + 1}]"
                out.append(f"\tv_mov_b64_e32 {tmp}, v[{d0}:{d0 + 1}]\n")
                out.append(m.group(1) + f"v[{d0}:{d0 + 1}], " + m.group(4).replace(f"v[{d0}:{d0 + 1}]", tmp) + "\n")
                n += 1
                continue
        if KERNEL.match(ln):
            cur = ln
        if inside:
            k = re.match(r"\s*\.amdhsa_next_free_vgpr (\d+)", ln)
            if k:
                nfv = (int(k.group(1)) + 1) // 2 * 2 + 2
                ln = ln.replace(k.group(1), str(nfv))
            k = re.match(r"\s*\.amdhsa_accum_offset (\d+)", ln)
            if k:
                ln = ln.replace(k.group(1), str((nfv + 3) // 4 * 4))
        out.append(ln)
    return out, n


def _mods(text, key, n, default):
    m = re.search(key + r":\[([01,]+)\]", text)
    v = [int(x) for x in m.group(1).split(",")] if m else []
    return v + [default] * (n - len(v))


def scalarise(ln, top):
    """One packed fp32 instruction -> the two scalar VOP3 instructions it stands for (results into two spare registers
    first, then moved: sources may overlap the destination).  Same arithmetic bit for bit (v_pk_fma_f32 is fused)."""
    m = re.match(r"\s*v_pk_(add|mul|fma)_f32\s+v\[(\d+):(\d+)\],\s*(.*)$", ln.split(";")[0].rstrip())
    kind, d0 = m.group(1), int(m.group(2))
    rest = m.group(4)
    mods = rest[rest.find(" op_sel"):] if " op_sel" in rest else (rest[rest.find(" neg_"):] if " neg_" in rest else "")
    ops = [o.strip() for o in (rest[:len(rest) - len(mods)] if mods else rest).split(",")]
    n = len(ops)
    sel, selhi = _mods(mods, "op_sel", n, 0), _mods(mods, "op_sel_hi", n, 1)
    neglo, neghi = _mods(mods, "neg_lo", n, 0), _mods(mods, "neg_hi", n, 0)

    def half(op, hi, neg):
        r = re.match(r"([vs])\[(\d+):(\d+)\]$", op)
        t = f"{r.group(1)}{int(r.group(2)) + hi}" if r else op       # (an inline constant feeds both halves)
        return ("-" if neg else "") + t
    mn = {"add": "v_add_f32_e64", "mul": "v_mul_f32_e64", "fma": "v_fma_f32"}[kind]
    lo = ", ".join(half(o, sel[i], neglo[i]) for i, o in enumerate(ops))
    hi = ", ".join(half(o, selhi[i], neghi[i]) for i, o in enumerate(ops))
    return [f"\t{mn} v{top}, {lo}\n", f"\t{mn} v{top + 1}, {hi}\n",
            f"\tv_mov_b32_e32 v{d0}, v{top}\n", f"\tv_mov_b32_e32 v{d0 + 1}, v{top + 1}\n"]


def edit_scalar(lines, want):
    """Packed instructions of the gradloc kernels for which ``want(kind, has_op_sel)`` holds become scalar pairs."""
    out, inside, n, top, cur, nfv = [], False, 0, None, None, None
    for ln in lines:
        if KERNEL.match(ln):
            inside, top, cur = True, None, ln
        elif inside and ln.startswith(".Lfunc_end"):
            inside = False
        m = PACKED.match(ln) if inside else None
        if m and want(m.group(1), ("B" if re.search(r"op_sel:\[[01,]*1", ln) else "C") if "op_sel" in ln else ""):
            if top is None:
                top = (TOPS[[k for k in TOPS if k in cur][0]] + 1) // 2 * 2
            out += scalarise(ln, top)
            n += 1
            continue
        if inside:
            k = re.match(r"\s*\.amdhsa_next_free_vgpr (\d+)", ln)
            if k:
                nfv = (int(k.group(1)) + 1) // 2 * 2 + 2
                ln = ln.replace(k.group(1), str(nfv))
            k = re.match(r"\s*\.amdhsa_accum_offset (\d+)", ln)
            if k:
                ln = ln.replace(k.group(1), str((nfv + 3) // 4 * 4))
        out.append(ln)
    return out, n


TOPS = {}       # kernel label -> first unused VGPR (its .amdhsa_next_free_vgpr)


def scan_tops(lines):
    cur = None
    for ln in lines:
        if KERNEL.match(ln):
            cur = ln
        k = re.match(r"\s*\.amdhsa_next_free_vgpr (\d+)", ln)
        if k and cur is not None:
            TOPS[cur] = int(k.group(1))
            cur = None


def chunks(lines):
    """{kernel symbol: (first, last)} line ranges that make up one kernel in the compiler's assembly: its .text section
    with the code, the kernel descriptor in .rodata and the trailing .set lines."""
    out, starts = {}, []
    for i, ln in enumerate(lines):
        m = re.match(r"\s*\.globl\s+(_ZN7bevmsda\w+)\s*$", ln)
        if m and not m.group(1).endswith(".kd"):
            j = i
            while not lines[j].lstrip().startswith(".section"):
                j -= 1
            starts.append((m.group(1), j))
    for name, j in starts:
        last = max(i for i, ln in enumerate(lines) if ln.lstrip().startswith(f".set {name}."))
        out[name] = (j, last)
    return out


def transplant(dst, src, pattern):
    """``dst`` with every kernel whose symbol matches ``pattern`` replaced by the same kernel of ``src``."""
    cd, cs = chunks(dst), chunks(src)
    names = [n for n in cd if re.search(pattern, n)]
    assert names and all(n in cs for n in names)
    out, pos = [], 0
    for n in sorted(names, key=lambda n: cd[n][0]):
        a, b = cd[n]
        out += dst[pos:a] + src[cs[n][0]:cs[n][1] + 1]
        pos = b + 1
    return out + dst[pos:], len(names)


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
    nos = open(noslp_s).readlines()
    for name, pat in (("gl_noslp", r"msda_gradloc_d32"), ("sort_noslp", r"msda_gradvalue_sort"),
                      ("glsort_noslp", r"msda_gradloc_d32|msda_gradvalue_sort")):
        ed, n = transplant(lines, nos, pat)
        p = os.path.join(outdir, name + ".s")
        open(p, "w").writelines(ed)
        assemble(p, os.path.join(outdir, name + ".hsaco"))
        os.remove(p)
        print(f"{name}: {n} kernels taken from the build without packed math, the rest of the unit keeps it")
    scan_tops(lines)
    for name, want in (("sc_all", lambda k, o: True), ("sc_add", lambda k, o: k == "add"),
                       ("sc_addsel", lambda k, o: k == "add" and o), ("sc_mulfma", lambda k, o: k != "add"),
                       ("sc_sel", lambda k, o: o), ("sc_nosel", lambda k, o: not o),
                       # B: op_sel:[..1..] — a LOW result lane reads the HIGH half of a source pair;
                       # C: op_sel_hi:[..0..] only — a HIGH result lane reads the LOW half (broadcast of the low half)
                       ("sc_selB", lambda k, o: o == "B"), ("sc_selC", lambda k, o: o == "C")):
        ed, n = edit_scalar(lines, want)
        p = os.path.join(outdir, name + ".s")
        open(p, "w").writelines(ed)
        assemble(p, os.path.join(outdir, name + ".hsaco"))
        print(f"{name}: {n} packed instructions of the gradloc kernels replaced by scalar pairs")
    ed, n = edit_inplace(lines)
    p = os.path.join(outdir, "inplace_tmp.s")
    open(p, "w").writelines(ed)
    assemble(p, os.path.join(outdir, "inplace_tmp.hsaco"))
    print(f"inplace_tmp: {n} in-place packed instructions with cross-selected halves now read a copy")
    packed = sum(1 for ln in edit(lines, lambda k: (["x"], []))[0] if ln == "\tx\n")
    print(f"slp: {packed} packed fp32 instructions inside msda_gradloc_d32_kernel<*>")
    assert not any(PACKED.match(ln) for ln in open(noslp_s)), "the no-SLP build must have no packed fp32 math"


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "tools", "probes", "pk_repro", "hsaco"))
