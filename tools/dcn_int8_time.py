#!/usr/bin/env python3
"""HIP-event medians of the DCNv2 flavours at the two ResNet-101 shapes: int8 LDS-DMA kernel (variant 0; 10 =
without SDWA converts), int8 register-staged fused kernel (8), int8 im2col + GEMM (6), fp16 default -- and of the
offset convolution's two variants (0 default, 1 rows-in-LDS)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402


def med(fn, iters=30, warm=5):
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
    return round(ts[len(ts) // 2] * 1e3, 1), round(ts[0] * 1e3, 1)


def main():
    lib = load_library()
    g = torch.Generator().manual_seed(0)
    for (B, C, H, W) in ((6, 256, 58, 100), (6, 512, 29, 50)):
        x = torch.randint(-127, 128, (B, C, H, W), generator=g, dtype=torch.int8).cuda()
        off = torch.randint(-127, 128, (B, 18, H, W), generator=g, dtype=torch.int8).cuda()
        mask = torch.randint(0, 128, (B, 9, H, W), generator=g, dtype=torch.int8).cuda()
        w = torch.randint(-127, 128, (C, C, 3, 3), generator=g, dtype=torch.int8).cuda()
        b = torch.zeros(C).cuda()
        for v in (0, 8, 6, 9, 5) + ((36, 21, 22, 24, 28) if C == 256 and os.environ.get('ABLATE') else ()):
            lib.bevops_mdconv_set_variant(v)
            try:
                m, mn = med(lambda: bev.modulated_deformable_conv2d_int8(x, off, mask, w, b, 0.02, 0.03, 1 / 127, 0.01, 0.05, 1, 1, 1, 1, 1))
                print(json.dumps({"op": "dcn_int8", "shape": [B, C, H, W], "variant": v, "us": m, "min_us": mn}), flush=True)
            finally:
                lib.bevops_mdconv_set_variant(0)
        xh = torch.randn(B, C, H, W, generator=g).half().cuda()
        oh = (torch.randn(B, 18, H, W, generator=g) * 2).half().cuda()
        mh = torch.rand(B, 9, H, W, generator=g).half().cuda()
        wh = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
        bh = torch.zeros(C).half().cuda()
        for v in (0, 7, 5):   # 7: wave halves in opposite phase order, 5: 128-pixel tiles whatever the tile count
            lib.bevops_mdconv_set_variant(v)
            try:
                m, mn = med(lambda: bev.modulated_deformable_conv2d(xh, oh, mh, wh, bh, 1, 1, 1, 1, 1))
                print(json.dumps({"op": "dcn_f16", "shape": [B, C, H, W], "variant": v, "us": m, "min_us": mn}), flush=True)
            finally:
                lib.bevops_mdconv_set_variant(0)
        xc = xh.contiguous(memory_format=torch.channels_last)
        w27 = (torch.randn(27, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
        b27 = torch.zeros(27).half().cuda()
        for v in (0, 1):
            lib.bevops_conv3x3_c32_set_variant(v)
            try:
                m, mn = med(lambda: bev.conv_offset_nhwc(xc, w27, b27))
                print(json.dumps({"op": "conv_offset_nhwc", "shape": [B, C, H, W], "variant": v, "us": m, "min_us": mn}), flush=True)
            finally:
                lib.bevops_conv3x3_c32_set_variant(0)


if __name__ == "__main__":
    main()
