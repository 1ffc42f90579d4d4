#!/usr/bin/env python3
"""How much of the fused DCN kernel's time is the partial last round?  Time the stage-3 channel
configuration at pixel counts around multiples of 512 tiles x 64 pixels."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev
from ops_timing import time_call
g = torch.Generator().manual_seed(0)
C = 256
for (B, H, W) in [(8, 64, 64), (8, 64, 68), (6, 58, 100), (8, 64, 128), (4, 64, 64), (2, 64, 64)]:
    x = torch.randn(B, C, H, W, generator=g).half().cuda()
    off = torch.randn(B, 18, H, W, generator=g).half().cuda()
    mask = torch.rand(B, 9, H, W, generator=g).half().cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / 48).half().cuda()
    b = torch.randn(C, generator=g).half().cuda()
    us = time_call(lambda: bev.modulated_deformable_conv2d(x, off, mask, w, b, 1, 1, 1, 1, 1))
    px = B * H * W
    print(json.dumps(dict(B=B, H=H, W=W, pixels=px, tiles=(px + 63) // 64, us=round(us, 1),
                          ns_per_pixel=round(us * 1e3 / px, 2), TFLOPs=round(2.0 * px * C * C * 9 / us / 1e6, 1))))
