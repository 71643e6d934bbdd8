"""copy_audit.py for the TRAINING step: which host statements launch the copy / fill / add / cat / mul kernels of one eager forward +
backward of the base encoder (autograd fast path)?  torch.profiler with stacks.  GPU box: python tools/copy_audit_train.py"""
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
g = torch.randn(1, 40000, 256, device=DEV)
qd, fd = q.clone().requires_grad_(True), f.clone().requires_grad_(True)


def step():
    enc.zero_grad(set_to_none=True)
    qd.grad = fd.grad = None
    enc(qd, fd, fd, **kw).backward(g)


for _ in range(2):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                            record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
names = ("aten::copy_", "aten::fill_", "aten::cat", "aten::zero_", "aten::clone", "aten::contiguous", "aten::index_select", "aten::stack",
         "aten::add", "aten::add_", "aten::mul", "aten::repeat", "aten::zeros", "aten::threshold_backward", "aten::_to_copy")
rows = collections.Counter()
for e in prof.events():
    if e.name in names:
        frame = next((s for s in e.stack if "bevformer_amd" in s and "profiler" not in s), e.stack[0] if e.stack else "(autograd engine)")
        rows[(e.name, str(e.input_shapes)[:64], frame.split("bevformer_amd/")[-1][:80])] += 1
for (name, shapes, frame), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:24s} {shapes:64s} {frame}")
