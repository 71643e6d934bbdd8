"""Builds ``lib/libbevmsda.so`` (HIP kernels + C ABI) for gfx950 with hipcc.

The library is compiled in-tree so that it travels with the repo snapshot to
the GPU box; there is no JIT and no torch C++ extension involved — the shared
object has a plain C ABI (include/bevmsda.h) and is loaded with ctypes.
Every ``csrc/*.hip`` file is one translation unit; they are compiled in
parallel into ``lib/obj/*.o`` (only the stale ones) and linked into one
shared object.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libbevmsda.so")
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "bevmsda.h")
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]


# per-source additions to FLAGS
EXTRA_FLAGS = {}
# sources that #include another source
INCLUDES = {"bevmsda_capi_backward.hip": ["bevmsda_capi.hip"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [PUBLIC_HEADER]
    return max(os.path.getmtime(f) for f in files if os.path.exists(f))


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _stale_objects():
    hm = _headers_mtime()
    out = []
    for s in sources():
        o = _obj(s)
        newest = max([hm] + [os.path.getmtime(os.path.join(CSRC, f)) for f in [s] + INCLUDES.get(s, [])])
        if not os.path.exists(o) or os.path.getmtime(o) < newest:
            out.append(s)
    return out


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    newest = max([_headers_mtime()] + [os.path.getmtime(os.path.join(CSRC, s)) for s in sources()])
    return os.path.getmtime(LIB_PATH) < newest


def build_library(force=False, verbose=False):
    """Compile every HIP source (stale ones only unless ``force``) and link one shared object.
    Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbevmsda.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = sources() if force else _stale_objects()

    def compile_one(src):
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", _obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(_obj(src) + ".tmp", _obj(src))

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"]
    cmd += [_obj(s) for s in sources()]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
