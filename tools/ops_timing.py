#!/usr/bin/env python3
"""Time the non-MSDA hot-path ops at their BEVFormer-base / BEVDet shapes (HIP events)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bevformer_tensorrt_amd as bev  # noqa: E402


def time_call(fn, iters=20, warm=3):
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
    return ts[len(ts) // 2] * 1e3


def main():  # noqa: C901
    g = torch.Generator().manual_seed(0)
    rows = []
    for dt in (torch.float16, torch.float32):
        es = 2 if dt == torch.float16 else 4
        # rotate: prev_bev [256,200,200]
        img = torch.randn(256, 200, 200, generator=g).to(dt).cuda()
        ang, ctr = torch.tensor(1.5).cuda(), torch.tensor([100.0, 100.0]).cuda()
        from bevformer_tensorrt_amd.utils import load_library
        for interp in ("nearest", "bilinear"):
            for variant, tag in ((0, ""), (1, "_per_lane_stores_r03")):     # LDS-transposed 16-byte stores vs rounds 1-3
                load_library().bevops_rotate_set_variant(variant)
                try:
                    us = time_call(lambda: bev.rotate(img, ang, ctr, interp))
                finally:
                    load_library().bevops_rotate_set_variant(0)
                rows.append(dict(op=f"rotate_{interp}{tag}", dtype=str(dt)[6:], us=round(us, 1),
                                 GBs=round(2 * img.numel() * es / us / 1e3, 1), frac_of_8TBs=round(2 * img.numel() * es / us / 8e6, 3)))
        # DCN stage 3 / 4 of R101 at base: 6 cams
        for name, (C, H, W) in {"dcn_s3": (256, 58, 100), "dcn_s4": (512, 29, 50)}.items():
            x = torch.randn(6, C, H, W, generator=g).to(dt).cuda()
            off = torch.randn(6, 18, H, W, generator=g).to(dt).cuda()
            mask = torch.rand(6, 9, H, W, generator=g).to(dt).cuda()
            w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dt).cuda()
            b = torch.randn(C, generator=g).to(dt).cuda()
            us = time_call(lambda: bev.modulated_deformable_conv2d(x, off, mask, w, b, 1, 1, 1, 1, 1))
            fl = 2.0 * 6 * H * W * C * C * 9
            rows.append(dict(op=name, dtype=str(dt)[6:], us=round(us, 1), TFLOPs=round(fl / us / 1e6, 1)))
        # grid_sampler reference test shape
        inp = torch.randn(8, 32, 100, 100, generator=g).to(dt).cuda()
        lin = torch.linspace(-15, 15, 1001)
        gy, gx = torch.meshgrid(lin, lin, indexing="ij")
        grid = torch.stack([gx, gy], 0)[None].repeat(8, 1, 1, 1).to(dt).cuda()
        us = time_call(lambda: bev.grid_sampler(inp, grid, "bilinear", "zeros", False), iters=10)
        byt = (inp.numel() + grid.numel() + 8 * 32 * 1001 * 1001) * es
        rows.append(dict(op="grid_sampler_2d_bilinear", dtype=str(dt)[6:], us=round(us, 1),
                         GBs=round(byt / us / 1e3, 1)))
    # bev_pool BEVDet-R50
    from util_bevpool import make_indices
    rd, rf, rb, ist, il = (torch.from_numpy(a).cuda() for a in make_indices(6, 59, 16, 44, 128, 128, keep=0.72))
    depth = torch.rand(6, 59, 16, 44, generator=g).half().cuda()
    feat = torch.randn(6, 16, 44, 64, generator=g).half().cuda()
    us = time_call(lambda: bev.bev_pool_v2(depth, feat, rd, rf, rb, ist, il, 128, 128))
    rows.append(dict(op="bev_pool_v2_r50", dtype="float16", us=round(us, 1), points=int(rd.numel())))
    for r in rows:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
