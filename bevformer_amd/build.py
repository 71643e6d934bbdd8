"""Builds ``lib/libbevmsda.so`` (HIP kernels + C ABI) for gfx950 with hipcc.

The library is compiled in-tree so that it travels with the repo snapshot to
the GPU box; there is no JIT and no torch C++ extension involved — the shared
object has a plain C ABI (include/bevmsda.h) and is loaded with ctypes.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbevmsda.so")
SOURCES = ["bevmsda_capi.hip"]
HEADERS = ["msda_kernels.h", os.path.join("..", "..", "include", "bevmsda.h")]
ARCH = "gfx950"


def _newest_source_mtime():
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    files += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    return max(os.path.getmtime(f) for f in files if os.path.exists(f))


def is_stale():
    return (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < _newest_source_mtime()


def build_library(force=False, verbose=False):
    """Compile every HIP source into one shared object.  Returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbevmsda.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", "-o", LIB_PATH + ".tmp"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
