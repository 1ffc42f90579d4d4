#!/usr/bin/env python3
"""Experimental head-major INT8 MSDA path (bevops_msda_set_variant(21)) vs the default INT8 kernel at the
base SCA / TSA shapes: bit-identity of the outputs (same per-item code, different addressing) and time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

lib = load_library()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def q8(x):
    s = float(x.abs().max()) / 127.0
    return torch.clamp(torch.round(x / s), -127, 127).to(torch.int8), s


def timed(fn, n=5):
    fn()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2] * 1e3


for name, (bs, levels, nq, P, ppg) in {"base_sca": (6, [[116, 200], [58, 100], [29, 50], [15, 25]], 40000, 8, 4),
                                       "base_tsa": (2, [[200, 200]], 40000, 4, 1)}.items():
    L, nk = len(levels), sum(h * w for h, w in levels)
    (v, sv), (o, so), (w, sw) = (q8(torch.randn(*s, device=dev, generator=g)) for s in
                                 ((bs, nk, 8, 32), (bs, nq, 8, L * P * 2), (bs, nq, 8, L * P)))
    sh = torch.tensor(levels, dtype=torch.int32, device=dev)
    row = {"shape": name}
    for rdt, tag in ((torch.float32, "s8w_f32ref"), (torch.float16, "u8w_f16ref")):
        ref = torch.rand(bs, nq, 1, 2 * ppg, device=dev, generator=g).to(rdt)
        call = lambda: bev.multi_scale_deformable_attn_int8(v, sh, ref, o, w, sv, so, sw, 0.02)
        lib.bevops_msda_set_variant(0)
        a, ta = call(), timed(call)
        lib.bevops_msda_set_variant(21)
        try:
            b, tb = call(), timed(call)
        finally:
            lib.bevops_msda_set_variant(0)
        row[tag] = {"identical": bool(torch.equal(a, b)), "default_us": round(ta, 1), "head_major_us": round(tb, 1)}
    print(json.dumps(row), flush=True)
