#!/usr/bin/env python3
"""Stage-3 / stage-4 calls of the DCNv2 pack's two kernels (offset convolution + deformable conv, NHWC)
back to back, for rocprofv3 --pmc passes: where do the image bytes of conv3x3_c32 come from?
The input is produced on the device right before each call (like conv1's output in the model)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402

g = torch.Generator().manual_seed(0)
for (B, C, H, W) in ((6, 256, 58, 100), (6, 512, 29, 50)):
    w27 = (torch.randn(27, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
    b27 = torch.zeros(27).half().cuda()
    wd = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
    bd = torch.zeros(C).half().cuda()
    src = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    for i in range(6):
        x = torch.relu(src * (1.0 + 0.01 * i))          # fresh producer output each time
        om = bev.conv_offset_nhwc(x, w27, b27)
        y = bev.modulated_deformable_conv2d_nhwc(x, None, None, wd, bd, 1, 1, 1, 1, 1, relu=True, offset_mask_nhwc=om)
    torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
