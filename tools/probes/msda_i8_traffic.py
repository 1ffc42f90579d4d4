#!/usr/bin/env python3
"""One int8 base SCA call per bevops_msda_set_variant value given on the command line (x127 flavour), for a
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass: the kernels of variant i run in the i-th call, in order."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from msda_sweep import SHAPES, gen  # noqa: E402

lib = load_library()
value, sh, ref, off, logit = gen(SHAPES["base_sca"], torch.float32, "uniform")[0]


def q(t):
    s = float(t.abs().max()) / 127.0
    return torch.clamp(torch.round(t / s), -127, 127).to(torch.int8), s


qv, s_v = q(value); qo, s_o = q(off); qw, s_w = q(logit)
for v in [int(a) for a in sys.argv[1:]]:
    lib.bevops_msda_set_variant(v)
    try:
        for _ in range(2):
            bev.multi_scale_deformable_attn_int8(qv, sh, ref, qo, qw, s_v, s_o, s_w, 0.02)
        torch.cuda.synchronize()
    finally:
        lib.bevops_msda_set_variant(0)
