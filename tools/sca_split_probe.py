#!/usr/bin/env python3
"""Level-class split of the fp16 base SCA call (round-3 review, item 2): does the call get faster as TWO kernels
that share a CU -- A: the big levels' samples through L2, no plane image in LDS; B: the staged levels' samples from
its LDS image -- than as one kernel whose in-order waves serialise both?  The A / B builds here are TIMING builds of
hm5 (bevops_msda_set_variant(1000 + 2048 | 4096 [+ 2]): each leaves the other class's taps out; no partial-sum
hand-off yet), launched alone and CONCURRENTLY on two streams.  If A || B is close to max(A, B) the correct pair
(+ ~0.25 GB of partial sums) is worth building; if it is close to A + B the CU's pipes are the limit, not the wave
order.  One JSON line per measurement; times include each launch's own re-layout pass (as the default call's does).
usage: sca_split_probe.py [uniform,rig] [rounds]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from msda_sweep import SHAPES, gen  # noqa: E402

A512, A1024, B512, B1024 = 1000 + 2048, 1000 + 2048 + 2, 1000 + 4096, 1000 + 4096 + 2


def main():
    dists = (sys.argv[1] if len(sys.argv) > 1 else "uniform,rig").split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = load_library()
    main_s = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def call(variant, args):
        lib.bevops_msda_set_variant(variant)
        try:
            return bev.multi_scale_deformable_attn(*args)
        finally:
            lib.bevops_msda_set_variant(0)

    def timed(launch, iters=10, warm=3):
        for _ in range(warm):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(iters):
            launch()
        e1.record(main_s)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    for dist in dists:
        args, byt = gen(SHAPES["base_sca"], torch.float16, dist)

        def single(v):
            return lambda: call(v, args)

        def pair(va, vb):
            def launch():
                s1.wait_stream(main_s)
                s2.wait_stream(main_s)
                with torch.cuda.stream(s1):
                    call(va, args)
                with torch.cuda.stream(s2):
                    call(vb, args)
                main_s.wait_stream(s1)
                main_s.wait_stream(s2)
            return launch

        cases = [("default (one kernel, pre-pass)", single(0)), ("default without pre-pass", single(1001)),
                 ("A alone, 512-thread blocks", single(A512)), ("A alone, 1024-thread blocks", single(A1024)),
                 ("B alone, 512-thread blocks", single(B512)), ("B alone, 1024-thread blocks", single(B1024)),
                 ("A512 || B512", pair(A512, B512)), ("A512 || B1024", pair(A512, B1024)),
                 ("A1024 || B512", pair(A1024, B512)), ("A1024 || B1024", pair(A1024, B1024)),
                 ("two default calls || (control: what two streams give a kernel that fills the CU)", pair(1001, 1001))]
        res = {name: [] for name, _ in cases}
        for _ in range(rounds):
            for name, launch in cases:
                res[name].append(round(timed(launch), 1))
        for name, _ in cases:
            med = sorted(res[name])[len(res[name]) // 2]
            print(json.dumps({"call": "base_sca fp16", "refs": dist, "case": name, "us": res[name], "us_med": med,
                              "frac_of_8TBs_if_it_were_the_call": round(byt / med / 8e6, 4)}), flush=True)


if __name__ == "__main__":
    main()
