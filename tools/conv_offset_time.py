#!/usr/bin/env python3
"""The DCNv2 pack's offset convolution (bevops_conv3x3_c32_forward_nhwc) at the model's shapes: the round-6 build with
the weights in registers and the image tile in LDS (default at Cin = 256; bit-identity with the tile kernel asserted)
against the tile kernel with three waves per 32-pixel tile (variant 3, round 5's default), one wave per tile (variant
2, rounds 1-4) and the rows-in-LDS kernel (variant 1), HIP-graph replay, interleaved.  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

lib = load_library()
SHAPES = [("base s3", 6, 256, 58, 100), ("base s4", 6, 512, 29, 50), ("small s3", 6, 256, 46, 80), ("small s4", 6, 512, 23, 40),
          ("base s3, one camera", 1, 256, 58, 100)]
for name, B, C, H, W in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(27, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
    b = torch.randn(27, generator=g).half().cuda()
    if "--once" in sys.argv:          # plain launches for rocprofv3 runs
        for v in (0, 3, 2, 1):
            lib.bevops_conv3x3_c32_set_variant(v)
            for _ in range(5):
                bev.conv_offset_nhwc(x, w, b)
        lib.bevops_conv3x3_c32_set_variant(0)
        torch.cuda.synchronize()
        continue
    lib.bevops_conv3x3_c32_set_variant(3)
    ref = bev.conv_offset_nhwc(x, w, b).clone()
    lib.bevops_conv3x3_c32_set_variant(0)
    assert torch.equal(bev.conv_offset_nhwc(x, w, b), ref)
    res = {"default": [], "split3": [], "one_wave": [], "rows": []}
    for _ in range(3):
        for key, v in (("default", 0), ("split3", 3), ("one_wave", 2), ("rows", 1)):
            lib.bevops_conv3x3_c32_set_variant(v)
            try:
                res[key].append(round(L.graph_time_us(lambda: bev.conv_offset_nhwc(x, w, b)), 2))
            finally:
                lib.bevops_conv3x3_c32_set_variant(0)
    byt = (B * H * W * (C + 32) + 27 * C * 9) * 2
    print(json.dumps({"shape": name, "B": B, "Cin": C, "H": H, "W": W, "us": {k: sorted(v)[1] for k, v in res.items()},
                      "byte_floor_us_at_8TBs": round(byt / 8e6, 2)}), flush=True)
