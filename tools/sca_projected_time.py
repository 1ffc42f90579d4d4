#!/usr/bin/env python3
"""value_proj + fused SCA sampling: separate projection + bevops_sca_forward vs the projected path
(bevops_value_proj_packed + bevops_sca_forward_prepacked), rig geometry, interleaved."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd import geometry as G  # noqa: E402
from msda_sweep import time_call  # noqa: E402

g = torch.Generator().manual_seed(0)
levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
nk = sum(h * w for h, w in levels)
nq, heads, embed = 40000, 8, 256
feats = (torch.randn(6, nk, embed, generator=g) * 0.5).half().cuda()
wgt = (torch.randn(embed, embed, generator=g) / 16).half().cuda()
bias = (torch.randn(embed, generator=g) * 0.1).half().cuda()
ref3d = G.reference_points_3d(200, 200, 8, 4, device="cpu")
cam, mask = G.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], G.synthetic_lidar2img((928, 1600)), (928, 1600))
ref = cam.reshape(6, nq, 1, 8).half().cuda()
bm = mask.reshape(6, nq, -1).any(-1).half().cuda()
off = torch.randn(1, nq, heads, 64, generator=g).half().cuda()
w = torch.randn(1, nq, heads, 32, generator=g).half().cuda()
sh = torch.tensor(levels, dtype=torch.int32)


def old():
    value = torch.addmm(bias, feats.view(-1, embed), wgt.t()).view(6, nk, heads, 32)
    return bev.spatial_cross_attention_sample(value, sh, ref, off, w, bm)


def new():
    return bev.spatial_cross_attention_projected(feats, wgt, bias, sh, ref, off, w, bm, heads)


res = {"separate": [], "projected": []}
for _ in range(3):
    res["separate"].append(round(time_call(old, iters=15, warm=4)[0], 1))
    res["projected"].append(round(time_call(new, iters=15, warm=4)[0], 1))
print(json.dumps(res))
