#!/usr/bin/env python3
"""ResNet stage-1 conv2 (64 -> 64, 3x3) under HIP-graph replay: tiled implicit GEMM, the LDS-resident kernel
(csrc/conv_halo.hip), the library convolution + epilogue pass; base / small / tiny shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_tensorrt_amd.functions import conv as CV  # noqa: E402
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402

g = torch.Generator().manual_seed(0)
for B, H, W, what in ((6, 232, 400, "base"), (6, 184, 320, "small"), (6, 120, 200, "tiny"), (6, 64, 176, "bevdet"), (1, 232, 400, "base, one camera")):
    x = torch.randn(B, 64, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, generator=g).half().cuda()
    assert torch.equal(CV.conv3x3_c64(x, w, b, True), CV.conv_nhwc(x, w, b, True))
    row = {"shape": [B, H, W], "what": what}
    for _ in range(2):
        for name, fn in (("tile", CV.conv_nhwc), ("halo", CV.conv3x3_c64), ("library", CV._library)):
            row.setdefault(name, []).append(round(graph_time_us(lambda: fn(x, w, b, True, None, 1), 4, 3), 1))
    print(json.dumps(row), flush=True)

