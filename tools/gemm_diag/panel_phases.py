"""Where a wavefront of the row-panel projection kernel spends its clocks (PANEL_CLK stamps of the diagnostic build,
tools/gemm_diag/panel_diag.hip): per phase the shader clocks summed over all wavefronts / the wavefront count.
usage (GPU box): python tools/gemm_diag/panel_phases.py"""
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
hdr = os.path.join(os.path.dirname(os.path.dirname(HERE)), "bevformer_amd", "csrc", "linear_panel.h")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed", src, "-o", so],
                   check=True)
lib = ctypes.CDLL(so)
lib.diag_panel_phases.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_long, ctypes.c_long] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
DEV = torch.device("cuda:0")
NAMES = ["panel DMA issue + set-up", "wait for the panel DMA", "split to planes + barrier", "k loops (MFMA phases)", "tile epilogues (bias, stores issued)",
         "-", "-", "wavefronts"]
for name, M, N, K, groups in (("sca_value_proj", 184950, 1536, 256, 6), ("tsa_value_proj", 80000, 1536, 256, 6)):
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.05
    b = torch.randn(N, device=DEV)
    y = torch.empty(groups, M, N // groups, device=DEV)
    blob = ops.panel_weight(w)
    st = torch.cuda.current_stream().cuda_stream
    for nprod in (3, 1):
        for shape in (2, 1):
            prof = torch.zeros(8, dtype=torch.int64, device=DEV)
            for _ in range(3):
                lib.diag_panel_phases(x.data_ptr(), K, blob.data_ptr(), blob.numel() * 2, b.data_ptr(), y.data_ptr(), N // groups, M, N, K,
                                      N // groups, nprod, shape, None, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.diag_panel_phases(x.data_ptr(), K, blob.data_ptr(), blob.numel() * 2, b.data_ptr(), y.data_ptr(), N // groups, M, N, K,
                                  N // groups, nprod, shape, prof.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            p = prof.tolist()
            waves = max(1, p[7])
            tot = sum(p[:5])
            print(f"{name} nprod {nprod} panel{64 * shape}: {e0.elapsed_time(e1) * 1e3:.0f} us with stamps; per wavefront: " +
                  ", ".join(f"{NAMES[i]} {p[i] / waves:.0f} ({100.0 * p[i] / tot:.0f} %)" for i in range(5)) + f"; sum {tot / waves:.0f} clocks, {waves} wavefronts")
