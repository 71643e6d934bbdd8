"""tools/copy_audit.py for one frame WITH history through PerceptionTransformer.get_bev_features (the caller of the encoder):
which host statements launch the copy / cat / fill kernels around the encoder.  GPU box: python tools/copy_audit_queue.py"""
import collections
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_amd  # noqa: E402
from bevformer_amd import synthetic as S  # noqa: E402
from bevformer_amd.history import BevHistory  # noqa: E402

DEV = torch.device("cuda:0")
tr = bevformer_amd.build_transformer(S.transformer_cfg("base")).eval()
tr.init_weights()
tr = tr.to(DEV)
mlvl, bq, tkw = S.make_transformer_inputs("base", seed=0, temporal=False, device=DEV)
tkw.pop("prev_bev")
metas = []
for i in range(3):
    m = copy.deepcopy(tkw["img_metas"])
    m[0]["scene_token"] = "s"
    m[0]["can_bus"][:3] = np.array([2.0 * (i + 1), 0.5 * (i + 1), 0.0])
    m[0]["can_bus"][-1] = 4.0 * (i + 1)
    metas.append(m)
rest = {k: v for k, v in tkw.items() if k != "img_metas"}
bev_fn = lambda f, m, p: tr.get_bev_features(f, bq, prev_bev=p, img_metas=m, **rest)
hist = BevHistory()
with torch.no_grad():
    hist.step(bev_fn, mlvl, metas[0])
    hist.step(bev_fn, mlvl, metas[1])
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                                record_shapes=True) as prof:
        hist.step(bev_fn, mlvl, metas[2])
        torch.cuda.synchronize()
rows = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::zero_", "aten::clone", "aten::contiguous", "aten::stack", "aten::to", "aten::add",
                  "aten::mul", "aten::index_select", "aten::zeros", "aten::_to_copy"):
        frames = [s for s in e.stack if "bevformer_amd" in s]
        frame = frames[0] if frames else (e.stack[0] if e.stack else "?")
        rows[(e.name, str(e.input_shapes)[:60], frame.split("bevformer_amd/")[-1][:80])] += 1
for (name, shapes, frame), n in sorted(rows.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{n:3d} x {name:16s} {shapes:60s} {frame}")
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60))
