#!/usr/bin/env python3
"""Run the base stage-3 DCN call a few times (for rocprofv3 passes). usage: dcn_one.py [variant] [iters] [int8]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd.utils import load_library
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(0)
C, H, W = 256, 58, 100
x = torch.randn(6, C, H, W, generator=g).half().cuda()
off = torch.randn(6, 18, H, W, generator=g).half().cuda()
mask = torch.rand(6, 9, H, W, generator=g).half().cuda()
w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
b = torch.randn(C, generator=g).half().cuda()
load_library().bevops_mdconv_set_variant(variant)
if len(sys.argv) > 3 and sys.argv[3] == "int8":
    xq = torch.randint(-127, 128, (6, C, H, W), generator=g, dtype=torch.int8).cuda()
    oq = torch.randint(-127, 128, (6, 18, H, W), generator=g, dtype=torch.int8).cuda()
    mq = torch.randint(0, 128, (6, 9, H, W), generator=g, dtype=torch.int8).cuda()
    wq = torch.randint(-127, 128, (C, C, 3, 3), generator=g, dtype=torch.int8).cuda()
    for _ in range(iters):
        bev.modulated_deformable_conv2d_int8(xq, oq, mq, wq, b.float(), 0.02, 0.03, 1 / 127, 0.01, 0.05, 1, 1, 1, 1, 1)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(iters):
    bev.modulated_deformable_conv2d(x, off, mask, w, b, 1, 1, 1, 1, 1)
torch.cuda.synchronize()
