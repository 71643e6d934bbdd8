"""Which host-side statements launch the small copy / fill / cat kernels of one encoder step?  torch.profiler with stacks over one eager
base frame; prints every aten::copy_ / fill_ / cat / zeros-like op with its shapes and the innermost bevformer_amd frame.
GPU box: python tools/copy_audit.py [R,G [layout]]   (R,G: the step of simulated rank R of a G-rank BEV-tiled job)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_amd  # noqa: E402
from bevformer_amd import synthetic as S  # noqa: E402

DEV = torch.device("cuda:0")
enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg("base")).eval().to(DEV)
q, f, kw = S.make_inputs("base", seed=0, temporal=True, device=DEV)
if len(sys.argv) > 1:
    from bevformer_amd import bev_tiling  # noqa: E402
    r, g = (int(v) for v in sys.argv[1].split(","))
    bev_tiling.enable_bev_tiling(enc, simulate=(r, g), layout=sys.argv[2] if len(sys.argv) > 2 else "sectors")
with torch.no_grad():
    for _ in range(2):
        enc(q, f, f, **kw)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                                record_shapes=True) as prof:
        enc(q, f, f, **kw)
        torch.cuda.synchronize()
rows = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::zero_", "aten::clone", "aten::contiguous", "aten::index_select", "aten::stack"):
        frame = next((s for s in e.stack if "bevformer_amd" in s and "profiler" not in s), e.stack[0] if e.stack else "?")
        rows[(e.name, str(e.input_shapes)[:70], frame.split("bevformer_amd/")[-1][:90])] += 1
for (name, shapes, frame), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:18s} {shapes:70s} {frame}")
print()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=70))
