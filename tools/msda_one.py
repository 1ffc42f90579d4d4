#!/usr/bin/env python3
"""Run one MSDA call shape/variant a few times (for rocprofv3 PMC passes).
usage: msda_one.py <shape> <variant> [dist] [iters]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd.utils import load_library
from msda_sweep import SHAPES, gen
LV = [[116, 200], [58, 100], [29, 50], [15, 25]]
SHAPES = dict(SHAPES, base_sca_l3=(6, LV[3:], 40000, 8, 4), base_sca_l23=(6, LV[2:], 40000, 8, 4),
              base_sca_l01=(6, LV[:2], 40000, 8, 4), base_sca_l0=(6, LV[:1], 40000, 8, 4))
shape, variant = sys.argv[1], int(sys.argv[2])
dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
args, _ = gen(SHAPES[shape], torch.float16, dist)
lib = load_library()
lib.bevops_msda_set_variant(variant)
for _ in range(iters):
    bev.multi_scale_deformable_attn(*args)
torch.cuda.synchronize()
