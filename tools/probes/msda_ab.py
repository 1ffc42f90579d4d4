#!/usr/bin/env python3
"""Interleaved A/B of bevops_msda_set_variant values on one MSDA call shape (fp16):
usage: msda_ab.py VARIANT_A VARIANT_B [shape=base_sca] [dist=uniform] [rounds=4]"""
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
    shape = sys.argv[3] if len(sys.argv) > 3 else "base_sca"
    dist = sys.argv[4] if len(sys.argv) > 4 else "uniform"
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    lib = load_library()
    args, byt = gen(SHAPES[shape], torch.float16, dist)
    res = {va: [], vb: []}
    for _ in range(rounds):
        for v in (va, vb):
            lib.bevops_msda_set_variant(v)
            try:
                res[v].append(round(time_call(lambda: bev.multi_scale_deformable_attn(*args), iters=15, warm=4)[0], 1))
            finally:
                lib.bevops_msda_set_variant(0)
    print(json.dumps({"call": shape, "refs": dist, "variants": [va, vb], "us_a": res[va], "us_b": res[vb]}), flush=True)


if __name__ == "__main__":
    main()
