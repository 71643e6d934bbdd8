"""Which part bounds the row-panel projection kernel?  Times linear_panel_kernel with parts switched off
(tools/gemm_diag/panel_diag.hip).  usage (GPU box): python tools/gemm_diag/panel_run.py"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bevformer_amd import ops  # noqa: E402

so = os.path.join(HERE, "libpaneldiag.so")
src = os.path.join(HERE, "panel_diag.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed", src, "-o", so],
                   check=True)
lib = ctypes.CDLL(so)
lib.diag_panel.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_long, ctypes.c_long] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
lib.diag_panel_policy.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_long, ctypes.c_long] + [ctypes.c_int] * 7 + [ctypes.c_void_p]
DEV = torch.device("cuda:0")
POLICIES = [(0, "weights 2 steps ahead (default)"), (3003, "weights 3 steps ahead"), (3004, "weights 4 steps ahead"),
            (3006, "weights 6 steps ahead")]
if "--paced" in sys.argv:
    POLICIES += [(200, "burst stores, nt fetch"), (1000, "paced stores, default fetch"), (1200, "paced stores, nt fetch")]
if "--policies" in sys.argv:
    POLICIES += [(2, "nt stores"), (16, "sc1 stores"), (18, "sc1 nt stores"), (216, "sc1 stores, nt fetch"), (202, "nt stores, nt fetch")]
MASKS = [(0, "full kernel"), (1, "no MFMA after step 0"), (2, "no stores"), (4, "weight fragments: steps 0-1 only"),
         (8, "activation fragments: steps 0-1 only"), (16, "no panel fetch / split"), (2 + 16, "no stores, no panel fetch"),
         (1 + 2, "no MFMA, no stores"), (2 + 4, "no stores, no weight loads"), (4 + 8 + 16, "MFMA + stores only"),
         (2 + 4 + 8 + 16, "MFMA only"), (1 + 4 + 8 + 16, "stores only")]


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


only = [a for a in sys.argv[1:] if not a.startswith('--')] or None
for name, M, N, K, groups in (("sca_value_proj", 184950, 1536, 256, 6), ("output_proj", 40000, 256, 256, 1),
                              ("ffn_fc1", 40000, 512, 256, 1)):
    if only and name not in only:
        continue
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.05
    y = torch.empty(groups, M, N // groups, device=DEV)
    blob = ops.panel_weight(w)
    st = torch.cuda.current_stream().cuda_stream
    print(f"{name}: M {M} N {N} K {K}  (fp32 in {M * K * 4 / 1e6:.0f} MB, out {M * N * 4 / 1e6:.0f} MB)")
    want = torch.nn.functional.linear(x[:4096], w).view(4096, groups, N // groups).transpose(0, 1)
    for nprod in (3, 1):
        for shape in (1, 2):
            for pol, label in POLICIES:
                call = lambda: lib.diag_panel_policy(x.data_ptr(), K, blob.data_ptr(), blob.numel() * 2, None, y.data_ptr(), N // groups,
                                                     M, N, K, N // groups if groups > 1 else 0, nprod, shape, 0, pol, st)
                y.zero_()
                t = timeit(call)
                err = (y[:, :4096] - want).abs().max().item()
                print(f"   nprod {nprod} shape {('64x64 ', '128x32', '64x32 ')[shape - 1]} policy {label:34s} {t:8.1f} us   max err vs torch {err:.2e}", flush=True)
    if "--masks" not in sys.argv:
        continue
    for nprod in (3, 1):
        for shape in (1, 2):
            for mask, label in MASKS:
                t = timeit(lambda: lib.diag_panel(x.data_ptr(), K, blob.data_ptr(), blob.numel() * 2, None, y.data_ptr(), N // groups,
                                                  M, N, K, N // groups if groups > 1 else 0, nprod, shape, mask, st))
                print(f"   nprod {nprod} panel{64 * shape:<4d} {label:42s} {t:8.1f} us", flush=True)
