#!/usr/bin/env python3
"""Phase probe of dcn_glds_f16_kernel<4> (bevops_mdconv_set_variant(116): a timing build whose waves overwrite the
head of one output pixel each with their s_memtime totals): where a wave of the ResNet-101 stage-3 call spends its
cycles -- blend (incl. the wait for its gathers), load issue, fragment reads + MFMAs, the trailing s_waitcnt, the barrier."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

lib = load_library()
g = torch.Generator().manual_seed(0)
B, C, H, W = 6, 256, 58, 100
x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
om = torch.randn(B, 32, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
wt = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
bs = torch.randn(C, generator=g).half().cuda()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 116      # 116: one order for all waves; 117: opposed halves
lib.bevops_mdconv_set_variant(variant)
for _ in range(3):
    out = bev.modulated_deformable_conv2d_nhwc(x, None, None, wt, bs, 1, 1, 1, 1, 1, True, om)
torch.cuda.synchronize()
lib.bevops_mdconv_set_variant(0)
rows = out.permute(0, 2, 3, 1).reshape(-1, C).contiguous().view(torch.int64).view(-1, C // 4)     # 64 int64 per pixel
tiles = 256                      # main tiles of 128 pixels
ncol = 6 if variant == 116 else 8
acc = torch.zeros(16, ncol, dtype=torch.float64)
n = 0
for t in range(0, tiles, 7):
    blk = rows[t * 128:t * 128 + 16, 8:8 + ncol].double().cpu()
    if (blk[:, ncol - 1] > 0).all() and (blk[:, ncol - 1] < 1e7).all():
        acc += blk
        n += 1
acc /= max(n, 1)
names = ["blend (+ wait for gathers)", "load issue (DMA + footprint + gathers)", "fragment reads + MFMAs", "s_waitcnt", "barrier", "loop total"]
if variant != 116:
    names = ["lower: blend", "DMA issue (+ lower: gathers)", "fragment reads + MFMAs", "upper: blend", "upper: gathers", "s_waitcnt", "barrier", "loop total"]
print(json.dumps({"tiles_sampled": n, "cycles_per_wave_mean_over_tiles": {nm: [round(float(v)) for v in acc[:, i]] for i, nm in enumerate(names)},
                  "mean_over_waves": {nm: round(float(acc[:, i].mean())) for i, nm in enumerate(names)}}), flush=True)
