"""Which resource bounds the projection kernel?  Times the production instantiation of
linear_splitbf16_kernel with parts of its K loop switched off (tools/gemm_diag/diag.hip).
usage (GPU box): python tools/gemm_diag/run.py      (builds tools/gemm_diag/libdiag.so when missing: needs hipcc)"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bevformer_amd import ops  # noqa: E402

so = os.path.join(HERE, "libdiag.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
                    os.path.join(HERE, "diag.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.diag_linear.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_int, ctypes.c_void_p]
DEV = torch.device("cuda:0")
MASKS = [(0, "full kernel"), (1, "no MFMA after chunk 0"), (2, "no stores"), (4, "A loads: chunk 0 only"),
         (8, "W copies: chunk 0 only"), (16, "no A split / LDS write after chunk 0"), (4 + 16, "no A loads + no A split"),
         (1 + 2, "no MFMA, no stores"), (4 + 8 + 16, "MFMA + fragment reads + stores only"),
         (1 + 4 + 8 + 16, "fragment reads + stores only"), (1 + 2 + 4 + 8 + 16, "fragment reads + barriers only")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, M, N, K, groups in (("ffn_fc1", 40000, 512, 256, 1), ("output_proj", 40000, 256, 256, 1),
                              ("ffn_fc2", 40000, 256, 512, 1), ("sca_value_proj", 184950, 1536, 256, 6)):
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.05
    y = torch.empty(groups, M, N // groups, device=DEV)
    blob = ops.packed_weight(w)
    st = torch.cuda.current_stream().cuda_stream
    print(f"{name}: M {M} N {N} K {K}  (fp32 in {M * K * 4 / 1e6:.0f} MB, out {M * N * 4 / 1e6:.0f} MB)")
    for nprod in (3, 1):
        for mask, label in MASKS:
            t = timeit(lambda: lib.diag_linear(x.data_ptr(), K, blob.data_ptr(), None, y.data_ptr(), N // groups, M, N, K,
                                               N // groups if groups > 1 else 0, nprod, mask, st))
            print(f"   nprod {nprod} {label:42s} {t:8.1f} us")
