import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_amd import ops, _lib
from gemm_small_m import timeit
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
w0, b0 = torch.randn(256, 256, device=DEV, generator=g) / 16, torch.randn(256, device=DEV, generator=g) * 0.1
fc1, fc2 = torch.nn.Linear(256, 512).to(DEV), torch.nn.Linear(512, 256).to(DEV)
n0, n1 = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
lib = _lib.load()
names = {0: "full", 1: "no DMA waits", 2: "no MFMA", 4: "no ring reads", 8: "no barriers", 16: "no DMA issue", 6: "no MFMA, no reads", 17: "no DMA at all",
         25: "no DMA, no barriers", 27: "no DMA, no barriers, no MFMA", 31: "nothing but the shell"}
for M in (2560, 32768):
    rows, res = torch.randn(M, 256, device=DEV, generator=g), torch.randn(M, 256, device=DEV, generator=g)
    ws = [ops.rowreg_weight(w0), ops.rowreg_weight(fc1.weight), ops.rowreg_weight(fc2.weight, kmajor=True)]
    y = torch.empty(M, 256, device=DEV)
    for diag, nm in names.items():
        desc = _lib.ChainDesc(M=M, ld_rows=256, ld_res=256, ld_y=256, C=256, F=512, precision=0, eps0=1e-5, eps1=1e-5)
        desc.reserved[1] = 3
        desc.reserved[2] = diag
        def call():
            rc = lib.bevmsda_proj_ffn_chain_f32(rows.data_ptr(), None, None, ws[0].data_ptr(), b0.data_ptr(), res.data_ptr(), n0.weight.data_ptr(), n0.bias.data_ptr(),
                                                ws[1].data_ptr(), fc1.bias.data_ptr(), ws[2].data_ptr(), fc2.bias.data_ptr(), n1.weight.data_ptr(), n1.bias.data_ptr(),
                                                ctypes.byref(desc), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        t = timeit(call, 10)[0]
        print(f"M {M:6d}  {nm:32s} {t:8.1f} us")
