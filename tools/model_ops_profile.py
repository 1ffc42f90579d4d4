#!/usr/bin/env python3
"""Operator-level (with input shapes) device-time breakdown of one steady-state eager frame of the
re-hosted model: which framework element-wise / copy ops are worth fusing next.
usage: model_ops_profile.py [model] [rows] [--int8]   (--int8: the PTQ build bench.py times, base only)"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

int8 = "--int8" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--int8"]
name = argv[0] if len(argv) > 0 else "base"
rows = int(argv[1]) if len(argv) > 1 else 70
dev, dtype = torch.device("cuda"), torch.float16
if int8:
    import bench
    runner = bench.ModelFrames(dev, "int8", 1, 0, None, None, graph=False).runner
else:
    model = B.BEVFormer(name).to(dev, dtype)
    runner = B.FrameRunner(model, dev, dtype, graph=False)
H, W = B.CONFIGS[name]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
img = torch.randn(1, 6, 3, H, W).to(dev, dtype)
for i in range(3):
    can = torch.zeros(18)
    can[0], can[-1] = 0.5 * i, 0.8 * i
    runner.step(img, can, l2i, "scene")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    can = torch.zeros(18)
    can[0], can[-1] = 2.0, 3.0
    runner.step(img, can, l2i, "scene")
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
limit = rows
lines = []
for e in ka:
    dev_us = getattr(e, "self_device_time_total", None)
    if dev_us is None:
        dev_us = e.self_cuda_time_total
    if dev_us <= 0 or not e.key.startswith(("aten::", "bevops", "miopen")):
        continue
    lines.append((dev_us, e.count, e.key, str(e.input_shapes)[:150]))
lines.sort(reverse=True)
tot = sum(r[0] for r in lines)
print(f"device time in framework / library operators: {tot / 1e3:.2f} ms over {sum(r[1] for r in lines)} calls")
for dev_us, n, key, shapes in lines[:limit]:
    print(f"{dev_us / 1e3:8.3f} ms  x{n:<4d} {key:<34s} {shapes}")
