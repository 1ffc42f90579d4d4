#!/usr/bin/env python3
"""Diagnostic: does a library dense layer give the same bits when its operands sit at other addresses?  (A heuristic that
picks the algorithm from the operands' alignment makes a frame irreproducible between two runs of one process: the
allocator hands out other addresses.)  Then three base frames twice with the framework's addmm excluded."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402

g = torch.Generator().manual_seed(0)
for M, N, K in ((40000, 512, 256), (40000, 256, 256), (139200, 256, 512)):
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    base = torch.empty(M * K + 4096, dtype=torch.half, device="cuda")
    x0 = torch.randn(M, K, generator=g).half().cuda()
    row = {}
    for name in ("torch", "blaslt", "tile"):
        fn = L._DENSE[name]
        ref = fn(x0, w, b, None, False).clone()
        same = {}
        for shift in (8, 64, 128, 1024):          # halves: 16 B, 128 B, 256 B, 2 KB
            x = base[shift:shift + M * K].view(M, K)
            x.copy_(x0)
            pad = torch.empty(shift * 3 + 8, dtype=torch.half, device="cuda")     # (moves the output's address too)
            same[shift] = bool(torch.equal(fn(x, w, b, None, False), ref))
            del pad
        row[name] = same
    print(json.dumps({"M": M, "N": N, "K": K, "same bits at shifted addresses": row}), flush=True)

from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402
from test_model_gpu import frames  # noqa: E402
dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer("base", seed=0).to(dev, dtype)
H, W = B.CONFIGS["base"]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)


def run():
    r = B.FrameRunner(model, dev, dtype)
    got = []
    for img, can, scene in frames((H, W), 3, dev, dtype):
        cls, crd = r.step(img, can, l2i, scene)
        got.append((r.prev_bev.clone(), cls.clone(), crd.clone()))
    return got


L._DENSE["torch"] = L._DENSE["blaslt"]      # the table's "torch" rows on the library call with the tuned, stored algorithm
a, b2 = run(), run()
print("default dispatch without addmm, two runs of three frames equal:",
      [bool(torch.equal(x, y)) for fa, fb in zip(a, b2) for x, y in zip(fa, fb)], flush=True)
