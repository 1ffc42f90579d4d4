#!/usr/bin/env python3
"""Times the hm5 builds (bevops_msda_set_variant(1000 + flags)) against hm3 on the base SCA call,
op-test reference points and the 6-camera rig geometry, interleaved; one JSON line per (refs, variant).
flags: 1 no pre-pass, 2 768 threads, 4 no big taps, 8 no staged taps, 16 operands once, 32 no store,
128 chunks of 2560 queries, 256 records through the LDS mailbox instead of DPP."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from msda_sweep import SHAPES, gen, time_call  # noqa: E402


def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,1000,1001,1256,1257,1128,1005,1009,1017,1033,1049,1057").split(",")]
    dists = (sys.argv[2] if len(sys.argv) > 2 else "uniform,rig").split(",")
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = load_library()
    shapes = dict(SHAPES)
    # same sample counts, smaller level 0 (plane 2.3 MB instead of 3.7): how much of the call is L2 capacity?
    shapes["sca_small_l0"] = (6, [[82, 141], [58, 100], [29, 50], [15, 25]], 40000, 8, 4)
    shapes["sca_tiny_l0"] = (6, [[58, 100], [58, 100], [29, 50], [15, 25]], 40000, 8, 4)
    shape = os.environ.get("SHAPE", "base_sca")
    for dist in dists:
        args, byt = gen(shapes[shape], torch.float16, dist)
        lib.bevops_msda_set_variant(16)
        want = bev.multi_scale_deformable_attn(*args).float()
        lib.bevops_msda_set_variant(0)
        res = {v: [] for v in variants}
        err = {}
        for r in range(rounds):
            for v in variants:
                lib.bevops_msda_set_variant(v)
                try:
                    if r == 0:
                        err[v] = round((bev.multi_scale_deformable_attn(*args).float() - want).abs().max().item(), 5)
                    res[v].append(round(time_call(lambda: bev.multi_scale_deformable_attn(*args), iters=12, warm=3)[0], 1))
                finally:
                    lib.bevops_msda_set_variant(0)
        for v in variants:
            med = sorted(res[v])[len(res[v]) // 2]
            print(json.dumps({"call": shape, "refs": dist, "variant": v, "us": res[v], "us_med": med,
                              "frac_of_8TBs": round(byt / med / 8e6, 4), "max_abs_vs_hm3": err[v]}), flush=True)


if __name__ == "__main__":
    main()
