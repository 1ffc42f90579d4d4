#!/usr/bin/env python3
"""The base TSA MSDA call (2 x 40 000 keys, 40 000 queries, 1 level x 4 points, fp16) on the reference points the MODEL
produces -- the regular BEV grid (encoder.py:170-195), neighbouring queries hit neighbouring pixels -- against the op
test's uniform-random points, for the head-major kernel (default at this map size: re-layout + msda_hm_kernel) and the
layout-preserving quad kernel (bevops_msda_set_variant(10)).  Graph-replay timing.  One JSON line per case."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

lib = load_library()
g = torch.Generator().manual_seed(0)
nq, heads, C, P = 40000, 8, 32, 4
value = torch.randn(2, nq, heads, C, generator=g).half().cuda()
off = torch.randn(2, nq, heads, P * 2, generator=g).half().cuda()
logit = torch.randn(2, nq, heads, P, generator=g).half().cuda()
shapes = torch.tensor([[200, 200]])
ys, xs = torch.meshgrid(torch.linspace(0.5, 199.5, 200) / 200, torch.linspace(0.5, 199.5, 200) / 200, indexing="ij")
grid = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, nq, 1, 2)
refs = {"bev grid (the model's)": torch.cat([grid + torch.tensor([0.01, -0.004]), grid]).half().cuda(),
        "uniform random (op test)": torch.rand(2, nq, 1, 2, generator=g).half().cuda()}
for rname, ref in refs.items():
    outs = {}
    for variant, vname in ((0, "default (head-major: re-layout + msda_hm_kernel)"), (10, "quad kernel, reference layout")):
        prev = lib.bevops_msda_set_variant(variant)
        try:
            fn = lambda: bev.multi_scale_deformable_attn(value, shapes, ref, off, logit)
            outs[variant] = fn().float()
            us = graph_time_us(fn, 8, 3)
        finally:
            lib.bevops_msda_set_variant(prev)
        print(json.dumps({"call": "base TSA MSDA fp16", "refs": rname, "kernel": vname, "us_graph_replay": round(us, 1)}), flush=True)
    print(json.dumps({"refs": rname, "max_abs_diff_between_kernels": round((outs[0] - outs[10]).abs().max().item(), 5)}), flush=True)
