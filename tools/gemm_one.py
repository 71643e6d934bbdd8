"""One projection shape in a loop (for rocprofv3 --pmc passes): python tools/gemm_one.py [sca|tsa] [kernel] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sca"
kern = sys.argv[2] if len(sys.argv) > 2 else "panel128"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
M = 184950 if which == "sca" else 80000
g = torch.Generator(device="cuda:0").manual_seed(0)
x = torch.randn(M, 256, device="cuda:0", generator=g)
w = torch.randn(1536, 256, device="cuda:0", generator=g) * 0.05
b = torch.randn(1536, device="cuda:0", generator=g)
ops.set_gemm_kernel(None if kern == "default" else kern)
with torch.no_grad():
    for _ in range(iters):
        y = ops.linear(x, w, b, groups=6)
torch.cuda.synchronize()
print(which, kern, float(y.float().abs().mean()))
