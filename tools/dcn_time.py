#!/usr/bin/env python3
"""HIP-event timing of the two BEVFormer-base DCN call shapes per kernel variant."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd.utils import load_library
from msda_sweep import time_call
lib = load_library()
g = torch.Generator().manual_seed(0)
for (C, H, W) in ((256, 58, 100), (512, 29, 50)):
    x = torch.randn(6, C, H, W, generator=g).half().cuda()
    off = torch.randn(6, 18, H, W, generator=g).half().cuda()
    mask = torch.rand(6, 9, H, W, generator=g).half().cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
    b = torch.randn(C, generator=g).half().cuda()
    for v in [int(a) for a in (sys.argv[1:] or ["0", "2", "1"])]:
        lib.bevops_mdconv_set_variant(v)
        med, mn = time_call(lambda: bev.modulated_deformable_conv2d(x, off, mask, w, b, 1, 1, 1, 1, 1))
        lib.bevops_mdconv_set_variant(0)
        fl = 2.0 * 6 * H * W * C * C * 9
        print(json.dumps(dict(shape=[6, C, H, W], variant=v, us_med=round(med, 1), us_min=round(mn, 1),
                              TFLOPs=round(fl / med / 1e6, 1))), flush=True)
