#!/usr/bin/env python3
"""BASELINE config 5 (BEVDet-R50) view-transform slice on the current GPU: depth_net -> softmax ->
bev_pool_v2 -> [1, 64, 128, 128] (det2trt/models/detector/bevdet.py:50-76) with the ranks of the
reference's geometry on the reference test's calibration; HIP-event medians, fp16 and fp32, plus the
INT8 bev_pool_v2 op alone.  One JSON line per measurement."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.bevdet import BEVDET_R50, LSSViewTransformer  # noqa: E402
from msda_sweep import time_call  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "bevdet_geometry.npz")))
t = lambda k: torch.from_numpy(g[k])
vt = LSSViewTransformer(**BEVDET_R50)
ranks = [r.cuda() for r in vt.get_bev_pool_input(t("sensor2ego"), None, t("cam2imgs"), t("post_rots"), t("post_trans"), t("bda"))]
x = torch.randn(6, 256, 16, 44)
for dtype in (torch.float16, torch.float32):
    m = vt.cuda().to(dtype)
    xs = x.cuda().to(dtype)
    med, mn = time_call(lambda: m.view_transform(xs, *ranks))
    y = m.depth_net(xs)
    depth = y[:, :59].softmax(dim=1).contiguous()
    feat = y[:, 59:123].permute(0, 2, 3, 1).contiguous()
    pmed, _ = time_call(lambda: bev.bev_pool_v2_2(depth, feat, ranks[1], ranks[2], ranks[0], ranks[3], ranks[4], 128, 128))
    byt = (depth.numel() + feat.numel() + 128 * 128 * 64) * depth.element_size() + sum(r.numel() for r in ranks) * 4
    print(json.dumps({"path": "bevdet_r50_view_transform", "dtype": str(dtype)[6:], "points": int(ranks[0].numel()),
                      "intervals": int(ranks[3].numel()), "slice_us": round(med, 1), "bev_pool_us": round(pmed, 1),
                      "bev_pool_GBps": round(byt / pmed / 1e3, 1)}), flush=True)
qd = torch.clamp(torch.round(depth.float() * 127), 0, 127).to(torch.int8)
qf = torch.clamp(torch.round(feat.float() / feat.float().abs().max() * 127), -127, 127).to(torch.int8)
med, _ = time_call(lambda: bev.bev_pool_v2_int8(qd, qf, ranks[1], ranks[2], ranks[0], ranks[3], ranks[4], 1 / 127, 0.02, 0.05, 128, 128))
print(json.dumps({"path": "bevdet_r50_bev_pool_v2", "dtype": "int8", "bev_pool_us": round(med, 1)}), flush=True)
