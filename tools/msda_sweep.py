#!/usr/bin/env python3
"""Time every MSDA call shape of the three BEVFormer models x dtype x kernel variant
on the current GPU (HIP events, many iterations) and print achieved algorithmic GB/s.
Tuning aid; bench.py is the contract benchmark."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

SHAPES = {
    "base_sca": (6, [[116, 200], [58, 100], [29, 50], [15, 25]], 40000, 8, 4),
    "base_tsa": (2, [[200, 200]], 40000, 4, 1),
    "base_dec": (1, [[200, 200]], 900, 4, 1),
    "small_sca": (6, [[23, 40]], 22500, 8, 4),
    "small_tsa": (2, [[150, 150]], 22500, 4, 1),
    "tiny_sca": (6, [[15, 25]], 2500, 8, 4),
    "tiny_tsa": (2, [[50, 50]], 2500, 4, 1),
}


def gen(shape, dtype, dist="uniform"):
    bs, levels, nq, P, ppg = shape
    heads, C = 8, 32
    g = torch.Generator().manual_seed(0)
    L = len(levels)
    nk = sum(h * w for h, w in levels)
    value = torch.randn(bs, nk, heads, C, generator=g)
    ref = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
    if dist == "rig":  # reference points from the model's own projection of BEV pillars
        from bevformer_tensorrt_amd import geometry as G
        bh = int(round(nq ** 0.5))
        img = {2500: (480, 800), 22500: (736, 1280), 40000: (928, 1600)}[nq]
        ref3d = G.reference_points_3d(bh, bh, 8, 4, device="cpu")
        cam, _ = G.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0],
                                  G.synthetic_lidar2img(img), img)
        ref = cam.reshape(6, nq, 1, 8).contiguous()     # spatial_cross_attention.py:254-262
    if dist == "oov":  # ~2/3 of (camera, query) pairs out of view, like a real rig
        shift = (torch.rand(bs, nq, 1, 1, generator=g) < 0.67).float() * 3.0
        ref = ref + shift
    off = torch.randn(bs, nq, heads, L * P * 2, generator=g)
    logit = torch.randn(bs, nq, heads, L * P, generator=g)
    sh = torch.tensor(levels, dtype=torch.int32)
    args = [value.to(dtype).cuda(), sh.cuda(), ref.to(dtype).cuda(), off.to(dtype).cuda(),
            logit.to(dtype).cuda()]
    es = torch.finfo(dtype).bits // 8
    byt = sum(a.numel() for a in (value, ref, off, logit)) * es + bs * nq * heads * C * es + 8 * L
    return args, byt


def time_call(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3  # median, min (us)


def main():
    lib = load_library()
    rows = []
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,10,11,15,16,17").split(",")]
    for name, shape in SHAPES.items():
        for dtype in [getattr(torch, d) for d in os.environ.get('DTYPES', 'float16').split(',')]:
            for dist in (("uniform", "rig") if name.endswith("sca") else ("uniform",)):
                args, byt = gen(shape, dtype, dist)
                for v in variants:
                    if v == 99 and name == "base_sca" and dtype == torch.float32:
                        pass
                    lib.bevops_msda_set_variant(v)
                    med, mn = time_call(lambda: bev.multi_scale_deformable_attn(*args),
                                        iters=10 if v == 99 else 30)
                    lib.bevops_msda_set_variant(0)
                    row = dict(shape=name, dtype=str(dtype).split(".")[-1], dist=dist, variant=v,
                               us_med=round(med, 1), us_min=round(mn, 1), MB=round(byt / 1e6, 1),
                               GBs=round(byt / med / 1e3, 1))
                    rows.append(row)
                    print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    main()
