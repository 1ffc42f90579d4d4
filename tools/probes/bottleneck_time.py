#!/usr/bin/env python3
"""A ResNet stage-1 bottleneck with identity shortcut (256 -> 64 -> 64 -> 256) under HIP-graph replay: the fused kernel
(csrc/bottleneck.hip) against what the model's dispatch runs for the three layers, and against the three hand-written
kernels in sequence; base / small / tiny / one-camera shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_tensorrt_amd.functions import conv as CV, linear as Ln  # noqa: E402
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402

g = torch.Generator().manual_seed(0)
for B, H, W, what in ((6, 232, 400, "base"), (6, 184, 320, "small"), (6, 120, 200, "tiny"), (1, 232, 400, "base, one camera")):
    x = torch.randn(B, 256, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(64, 256, 1, 1, generator=g) / 16).half().cuda()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().cuda()
    w3 = (torch.randn(256, 64, 1, 1, generator=g) / 8).half().cuda()
    b1, b2, b3 = (torch.randn(c, generator=g).half().cuda() for c in (64, 64, 256))
    rows = x.permute(0, 2, 3, 1).reshape(-1, 256)

    def chain(dense):
        y1 = dense(rows, w1.view(64, 256), b1, None, True).view(B, H, W, 64).permute(0, 3, 1, 2)
        y2 = CV.conv3x3_auto(y1, w2, b2, True)
        return dense(y2.permute(0, 2, 3, 1).reshape(-1, 64), w3.view(256, 64), b3, rows, True)

    fused = lambda: CV.bottleneck_c256_64(x, w1, b1, w2, b2, w3, b3)
    ref = chain(Ln.tile_gemm).view(B, H, W, 256).permute(0, 3, 1, 2)
    assert torch.equal(fused(), ref)
    row = {"shape": [B, H, W], "what": what}
    for _ in range(2):
        for name, fn in (("dispatch_three_launches", lambda: chain(Ln.dense_auto)), ("own_three_launches", lambda: chain(Ln.tile_gemm)),
                         ("fused", fused)):
            row.setdefault(name, []).append(round(graph_time_us(fn, 4, 3), 1))
    print(json.dumps(row), flush=True)
