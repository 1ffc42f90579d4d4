#!/usr/bin/env python3
"""Diagnostic: every dense-layer candidate (tsgemm / tile / blaslt / torch) on the encoder's layer shapes, five
evaluations of the same operands each: is the candidate run-to-run deterministic, and what does it cost (HIP-graph
replay)?  Then three base frames twice under the reproducible dispatch (own kernels only): bit-equal or not."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as _lib  # noqa: E402

g = torch.Generator().manual_seed(0)
for M, N, K, res in ((40000, 512, 256, False), (40000, 256, 256, False), (40000, 256, 256, True), (40000, 192, 256, False),
                     (40000, 512, 256, True), (40000, 256, 512, True)):
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if res else None
    row = {}
    for name, fn in L._DENSE.items():
        try:
            outs = [fn(x, w, b, r, False).clone() for _ in range(5)]
        except _lib.BevopsError:
            continue
        except Exception as exc:      # noqa: BLE001
            row[name] = repr(exc)[:60]
            continue
        torch.cuda.synchronize()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        row[name] = {"deterministic": same, "us": round(L.graph_time_us(lambda: fn(x, w, b, r, False), 4, 3), 1)}
    print(json.dumps({"M": M, "N": N, "K": K, "residual": res, "candidates": row}), flush=True)

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


L.DETERMINISTIC["enabled"] = True
a, b2 = run(), run()
print("reproducible dispatch, two runs of three frames equal:",
      [bool(torch.equal(x, y)) for fa, fb in zip(a, b2) for x, y in zip(fa, fb)], flush=True)
L.DETERMINISTIC["enabled"] = False
