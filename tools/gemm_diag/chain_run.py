"""Where does a workgroup of the row-chain kernel spend its time?  Phase clocks (lane 0 of wavefront 0, summed over
workgroups) of csrc/linear_chain.h at the base shape (40,000 rows, two-row gather).  GPU box: python tools/gemm_diag/chain_run.py"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bevformer_amd import ops  # noqa: E402

so, src = os.path.join(HERE, "libchaindiag.so"), os.path.join(HERE, "chain_diag.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
                    "-I" + os.path.join(os.path.dirname(os.path.dirname(HERE)), "bevformer_amd", "csrc"), src, "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.diag_rowreg_pack.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def rowreg_weight(w, kmajor=False):
    """Weight image of the retired register-resident kernel (tools/experimental/linear_rowreg.h)."""
    blob = torch.empty(w.shape[0] * w.shape[1] * 2, dtype=torch.int16, device=w.device)
    lib.diag_rowreg_pack(w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], int(kmajor), blob.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return blob


lib.diag_chain.argtypes = [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_void_p] * 13 + [ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
DEV = torch.device("cuda:0")
SHAPE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
R = M + 5960
g = torch.Generator().manual_seed(0)
rows = torch.randn(R, 256, generator=g).to(DEV)
idx = torch.full((M, 2), -1, dtype=torch.int32)
idx[:, 0] = torch.arange(M, dtype=torch.int32)
idx[:min(R - M, M), 1] = torch.arange(M, M + min(R - M, M), dtype=torch.int32)
idx = idx.to(DEV)
scale = (1.0 / (idx >= 0).sum(1).clamp(min=1).float()).contiguous()
w0, w1, w2 = (torch.randn(256, 256, generator=g) / 16).to(DEV), (torch.randn(512, 256, generator=g) / 16).to(DEV), (torch.randn(256, 512, generator=g) / 22).to(DEV)
b0, b1, b2 = torch.randn(256, device=DEV) * 0.1, torch.randn(512, device=DEV) * 0.1, torch.randn(256, device=DEV) * 0.1
res = torch.randn(M, 256, generator=g).to(DEV)
ga, be = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
if SHAPE == 3:
    p0, p1, p2 = rowreg_weight(w0), rowreg_weight(w1), rowreg_weight(w2, kmajor=True)
else:
    p0, p1, p2 = ops.panel_weight(w0), ops.panel_weight(w1), ops.panel_weight(w2)
y = torch.empty(M, 256, device=DEV)
prof = torch.zeros(12, dtype=torch.int64, device=DEV)
st = torch.cuda.current_stream().cuda_stream
call = lambda: lib.diag_chain(rows.data_ptr(), 256, idx.data_ptr(), scale.data_ptr(), p0.data_ptr(), b0.data_ptr(), res.data_ptr(),
                              ga.data_ptr(), be.data_ptr(), p1.data_ptr(), b1.data_ptr(), p2.data_ptr(), b2.data_ptr(), ga.data_ptr(),
                              be.data_ptr(), M, y.data_ptr(), prof.data_ptr(), SHAPE, st)
if SHAPE == 3:
    # launch time of the kernel cut short after each phase (differences = the phases)
    import statistics
    names3 = ["constants, row loads, split", "+ GEMM 0 (8 chunks)", "+ LayerNorm 0 + split", "+ FFN (32 chunks)", "+ LayerNorm 1", "+ stores = the kernel"]
    prev = 0.0
    for k, nm in enumerate(names3):
        stop = k + 1 if k < 5 else 0
        c = lambda: lib.diag_chain(rows.data_ptr(), 256, idx.data_ptr(), scale.data_ptr(), p0.data_ptr(), b0.data_ptr(), res.data_ptr(),
                                   ga.data_ptr(), be.data_ptr(), p1.data_ptr(), b1.data_ptr(), p2.data_ptr(), b2.data_ptr(), ga.data_ptr(),
                                   be.data_ptr(), M, y.data_ptr(), prof.data_ptr(), 30 + stop, st)
        for _ in range(3):
            c()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(10):
                c()
            b_.record()
            torch.cuda.synchronize()
            ts.append(a_.elapsed_time(b_) * 100)
        t = statistics.median(ts)
        print(f"   {nm:36s} {t:8.1f} us   (+{t - prev:6.1f})")
        prev = t
    sys.exit(0)
for _ in range(3):
    call()
torch.cuda.synchronize()
prof.zero_()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
N = 10
for _ in range(N):
    call()
b.record()
torch.cuda.synchronize()
names3 = ["constants, row loads, split", "GEMM 0 (8 chunks)", "LayerNorm 0 + split", "FFN hidden-tile GEMMs (16 chunks)", "FFN bias + ReLU + split (x 16)",
          "FFN output GEMMs (16 chunks)", "LayerNorm 1", "stores"]
names = ["panel fetch + split", "GEMM 0 (out_proj)", "bias + res + LayerNorm 0 + planes", "GEMM 1 half 0", "bias + ReLU + planes",
         "GEMM 2 half 0", "GEMM 1 half 1", "bias + ReLU + planes", "GEMM 2 half 1", "bias + res + LayerNorm 1", "stores"]
p = prof.cpu().tolist()
nb = (M + 127) // 128 if SHAPE == 3 else (M + 63) // 64 if SHAPE == 1 else (M + 31) // 32
if SHAPE == 3:
    names = names3
print(f"shape {SHAPE}: launch {a.elapsed_time(b) / N * 1e3:.1f} us (with the clock stamps); {nb} workgroups; cycles per workgroup and phase (mean):")
tot = sum(p[:len(names)])
for n, c in zip(names, p):
    print(f"   {n:36s} {c / N / nb:9.0f} clk  {100.0 * c / tot:5.1f} %")
print(f"   {'sum':36s} {tot / N / nb:9.0f} clk  (MFMA floor of a workgroup: 5 x 16 steps x 6 MFMA x 32 clk x 2 waves / SIMD = 30,720 clk)")
