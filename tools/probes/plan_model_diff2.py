#!/usr/bin/env python3
"""Diagnostic: three base frames under several switch settings, each run TWICE with the plan and once without: which
frames / tensors differ, and whether two identical runs agree at all."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402
from bevformer_tensorrt_amd.functions import conv as Cv  # noqa: E402
from bevformer_tensorrt_amd.functions import spatial_cross_attention as S  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as L  # noqa: E402
from test_model_gpu import frames  # noqa: E402

dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer("base", seed=0).to(dev, dtype)
H, W = B.CONFIGS["base"]["image"]
l2i_a = G.synthetic_lidar2img((H, W)).to(dev)
l2i_b = l2i_a.clone()
l2i_b[:, :, 0, 3] += 3.0
h = L.load_library()


def run(planned, graph):
    S.PLANNED["enabled"] = planned
    try:
        r = B.FrameRunner(model, dev, dtype, graph=graph)
        got = []
        for i, (img, can, scene) in enumerate(frames((H, W), 3, dev, dtype)):
            cls, crd = r.step(img, can, l2i_a if i < 2 else l2i_b, scene)
            got.append((r.prev_bev.clone(), cls.clone(), crd.clone()))
        return got
    finally:
        S.PLANNED["enabled"] = True


def cmp(a, b):
    return ["%s%d:%s" % (n, f, "=" if torch.equal(x, y) else "%.1e" % float((x.float() - y.float()).abs().max()))
            for f, (fa, fb) in enumerate(zip(a, b)) for n, x, y in zip("bcd", fa, fb)]


for name, stem, direct, graph in (("default", True, 3012, True), ("eager", True, 3012, False),
                                  ("library stem", False, 3012, True), ("scratch", True, 3013, True)):
    Cv.STEM_FUSED["enabled"] = stem
    h.bevops_msda_set_variant(direct)
    p1, p2, n1, n2 = run(True, graph), run(True, graph), run(False, graph), run(False, graph)
    print(name, "| plan vs plan", cmp(p1, p2), "| none vs none", cmp(n1, n2), "| plan vs none", cmp(p1, n1), flush=True)
