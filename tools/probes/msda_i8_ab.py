#!/usr/bin/env python3
"""Interleaved A/B of bevops_msda_set_variant values on the base SCA call in int8 (both flavours), with a
bit-comparison of variant B against variant A first.  usage: msda_i8_ab.py VARIANT_A VARIANT_B [rounds=4]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/ (msda_sweep)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from msda_sweep import SHAPES, gen, time_call  # noqa: E402


def main():
    va, vb = int(sys.argv[1]), int(sys.argv[2])
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lib = load_library()
    value, sh, ref, off, logit = gen(SHAPES["base_sca"], torch.float32, "uniform")[0]

    def q(t):
        s = float(t.abs().max()) / 127.0
        return torch.clamp(torch.round(t / s), -127, 127).to(torch.int8), s
    qv, s_v = q(value); qo, s_o = q(off); qw, s_w = q(logit)
    for rdt in (torch.float32, torch.float16):
        r = ref.to(rdt)
        call = lambda: bev.multi_scale_deformable_attn_int8(qv, sh, r, qo, qw, s_v, s_o, s_w, 0.02)  # noqa: E731
        outs = {}
        for v in (va, vb):
            lib.bevops_msda_set_variant(v)
            try:
                outs[v] = call().clone()
            finally:
                lib.bevops_msda_set_variant(0)
        torch.cuda.synchronize()
        diff = (outs[va].int() - outs[vb].int()).abs()
        res = {va: [], vb: []}
        for _ in range(rounds):
            for v in (va, vb):
                lib.bevops_msda_set_variant(v)
                try:
                    res[v].append(round(time_call(call, iters=15, warm=4)[0], 1))
                finally:
                    lib.bevops_msda_set_variant(0)
        print(json.dumps({"call": "base_sca", "dtype": "i8", "ref": str(rdt)[6:], "variants": [va, vb],
                          "mismatch": int((diff > 0).sum()), "max_abs": int(diff.max()),
                          "us_a": res[va], "us_b": res[vb]}), flush=True)


if __name__ == "__main__":
    main()
