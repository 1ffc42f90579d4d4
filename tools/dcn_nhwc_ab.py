#!/usr/bin/env python3
"""The channels-last DCNv2 entry the frame calls (bevops_mdconv_forward_nhwc: dcn_glds_f16_kernel + tail finish) at the
two BEVFormer-base shapes (and one camera of the first), per wave-order build of the kernel, under HIP-graph replay,
interleaved: variant 0 = the default (round 6: the lower half of the waves issues all the weight DMA, the upper half runs
its matrix segment first), 13 = all 16 waves in one order (rounds 2-5), 7 = upper half rotated by half an iteration
(round 2).  Results must be bit-identical.
usage: dcn_nhwc_ab.py [variant ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

lib = load_library()
once = 0
if "--once" in sys.argv:          # --once K: K plain launches per variant of the first shape (rocprofv3 runs)
    i = sys.argv.index("--once")
    once = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
variants = [int(a) for a in sys.argv[1:]] or [0, 13, 7]
g = torch.Generator().manual_seed(0)
for (B, C, H, W) in ((6, 256, 58, 100), (6, 512, 29, 50), (1, 256, 58, 100)):
    x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    om = torch.randn(B, 32, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
    bs = torch.randn(C, generator=g).half().cuda()

    def call():
        return bev.modulated_deformable_conv2d_nhwc(x, None, None, wt, bs, 1, 1, 1, 1, 1, True, om)

    if once:
        for v in variants:
            lib.bevops_mdconv_set_variant(v)
            for _ in range(once):
                call()
        torch.cuda.synchronize()
        lib.bevops_mdconv_set_variant(0)
        break
    lib.bevops_mdconv_set_variant(0)
    want = call()
    res = {v: [] for v in variants}
    same = {}
    for v in variants:
        lib.bevops_mdconv_set_variant(v)
        same[v] = bool(torch.equal(call(), want))
    for _ in range(3):
        for v in variants:
            lib.bevops_mdconv_set_variant(v)
            res[v].append(round(graph_time_us(call), 2))
    lib.bevops_mdconv_set_variant(0)
    fl = 2.0 * B * H * W * C * C * 9
    med = {v: sorted(t)[1] for v, t in res.items()}
    print(json.dumps({"shape": [B, C, H, W], "us": med, "identical_to_variant_0": same,
                      "frac_of_2.5PF": {v: round(fl / t / 1e6 / 2500.0, 4) for v, t in med.items()}}), flush=True)
