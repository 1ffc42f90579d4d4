#!/usr/bin/env python3
"""The plain 3x3 convolutions of BEVFormer-base's backbone / neck: implicit GEMM on the tiled MFMA skeleton
(bevops_conv3x3_tile_f16, shift + ReLU in the epilogue) vs the library convolution + epilogue pass -- what
functions/conv.py's conv3x3_auto measures and picks."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions import conv as Cv  # noqa: E402

SHAPES = [("s1.conv2", 6, 64, 232, 400, 64, True), ("s2.conv2", 6, 128, 116, 200, 128, True),
          ("fpn.out0", 6, 256, 116, 200, 256, False), ("fpn.out1", 6, 256, 58, 100, 256, False),
          ("fpn.out2", 6, 256, 29, 50, 256, False)]
for name, B, C, H, W, Cout, relu in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, generator=g).half().cuda()
    bev.conv3x3_auto(x, w, b, relu)
    key, times = Cv.CONV_LOG[-1]
    fl = 2.0 * B * H * W * 9 * C * Cout
    print(json.dumps({"layer": name, "shape": [B, C, H, W, Cout], "us": times, "pick": Cv._CHOICE[key],
                      "TFLOPs_tile": round(fl / times["tile"] / 1e6, 1)}), flush=True)
