#!/usr/bin/env python3
"""Which framework (aten) kernels does an eager base frame still launch, and from which line of bevformer.py?
torch.profiler with stacks over a few steady-state frames; one line per (aten op, call site): launches per frame and
device microseconds per frame, largest first.  usage: glue_ops.py [base] [--int8]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "base")
dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer(name, seed=0).to(dev, dtype)
H, W = B.CONFIGS[name]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
r = B.FrameRunner(model, dev, dtype)
can = torch.zeros(18)
for i in range(3):
    can[0], can[-1] = 0.5 * i, 0.8 * i
    r.step(img, can, l2i, "s")
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(N):
        can[0], can[-1] = 0.5 * (i + 3), 0.8 * (i + 3)
        r.step(img, can, l2i, "s")
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
        continue
    site = "?"
    for fr in (e.stack or []):
        if "bevformer_tensorrt_amd" in fr and "/functions/" not in fr or "geometry.py" in fr:
            site = fr.split("bevformer_tensorrt_amd/")[-1].split(",")[0][:60] if "bevformer_tensorrt_amd" in fr else fr[:60]
            break
    k = (e.name, site)
    agg[k][0] += 1
    agg[k][1] += e.device_time_total
tot = sum(v[1] for v in agg.values()) / N
print(f"aten kernels per frame: {sum(v[0] for v in agg.values()) / N:.0f} launches, {tot:.0f} us on the device")
for (op, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us / N:8.1f} us  x{n / N:5.1f}  {op:32s} {site}")
