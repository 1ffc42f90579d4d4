#!/usr/bin/env python3
"""Operator-level (with input shapes) device-time breakdown of one steady-state eager frame of the
re-hosted model: which framework element-wise / copy ops are worth fusing next.
usage: model_ops_profile.py [model] [rows]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 70
dev, dtype = torch.device("cuda"), torch.float16
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
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=rows,
                                                          max_name_column_width=40, max_shapes_column_width=70))
