#!/usr/bin/env python3
"""Where do hm4-int8 and the layout-preserving int8 kernel differ?  Prints mismatch counts for the
whole call and with the softmax mass forced onto one pyramid level at a time, plus examples."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from test_msda_int8_gpu import make, quantize  # noqa: E402
import oracle  # noqa: E402

lib = load_library()
BASE = [[116, 200], [58, 100], [29, 50], [15, 25]]


def run(args, scales, v):
    lib.bevops_msda_set_variant(v)
    try:
        o = bev.multi_scale_deformable_attn_int8(*args, *scales)
        torch.cuda.synchronize()
    finally:
        lib.bevops_msda_set_variant(0)
    return o


for rdt in (torch.float32, torch.float16):
    value, sh, ref, off, logit = make((6, BASE, 4096, 8, 4))
    qv, s_v = quantize(value); qo, s_o = quantize(off); qw, s_w = quantize(logit)
    for only in (None, 0, 1, 2, 3):
        w = qw.clone()
        if only is not None:
            w = w.view(6, 4096, 8, 4, 8)
            keep = w[:, :, :, only].clone()
            w[:] = -127
            w[:, :, :, only] = keep
            w = w.view(6, 4096, 8, 32)
        args = (qv.cuda(), sh.cuda(), ref.to(rdt).cuda(), qo.cuda(), w.cuda())
        scales = (s_v, s_o, s_w, 0.02)
        a = run(args, scales, 17).cpu().numpy().astype(np.int32)
        b = run(args, scales, 10).cpu().numpy().astype(np.int32)
        g = run(args, scales, 99).cpu().numpy().astype(np.int32)
        want = oracle.msda_s8(qv.numpy(), s_v, sh.numpy(), ref.to(rdt).float().numpy(), qo.numpy(), s_o, w.numpy(), s_w,
                              0.02, u8_weights=(rdt == torch.float16)).astype(np.int32)
        print(f"ref={str(rdt)[6:]} level={only}: hm4!=quad {int((a != b).sum())}  hm4!=generic {int((a != g).sum())} "
              f"quad!=generic {int((b != g).sum())}  hm4!=oracle {int((a != want).sum())} quad!=oracle {int((b != want).sum())} "
              f"generic!=oracle {int((g != want).sum())} of {a.size}; max|hm4-quad| {int(np.abs(a - b).max())}")
        idx = np.argwhere(a != b)[:6]
        for i in idx:
            i = tuple(i)
            print("    at (b,q,h,c)", i, "hm4", a[i], "quad", b[i], "generic", g[i], "oracle", want[i])
