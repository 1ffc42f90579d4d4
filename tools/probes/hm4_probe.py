#!/usr/bin/env python3
"""A/B timing of the head-major MSDA generations on the current GPU (HIP events, median of 30):
fp16 base SCA (uniform refs of the op test and the model's own camera geometry) on hm3 (16) / hm4 (17,
chunk sizes 170+k), base / small TSA on the old hm kernel (0) vs hm4 (17), and both int8 flavours
on the layout-preserving kernel (10) vs hm4.  One JSON line per measurement."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/ (msda_sweep)
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from msda_sweep import SHAPES, gen, time_call  # noqa: E402


def ablations(lib):
    """Time the base SCA call with parts of the hm4 kernel compiled out (variants 200 + mask; results
    are wrong by construction): 1 no L1/L2 taps, 2 no LDS taps, 4 no operand requests in the loop,
    8 no front end in the loop, 16 no store."""
    args, byt = gen(SHAPES["base_sca"], torch.float16, "uniform")
    value, sh, ref, off, logit = gen(SHAPES["base_sca"], torch.float32, "uniform")[0]

    def q(t):
        s = float(t.abs().max()) / 127.0
        return torch.clamp(torch.round(t / s), -127, 127).to(torch.int8), s
    qv, s_v = q(value); qo, s_o = q(off); qw, s_w = q(logit)
    for mask in [int(m) for m in os.environ.get("MASKS", "0,1,2,3,4,8,12,15,16,19,31,11,7").split(",")]:
        for dt in ("f16", "i8"):
            lib.bevops_msda_set_variant(200 + mask if mask else 17)
            try:
                if dt == "f16":
                    med, mn = time_call(lambda: bev.multi_scale_deformable_attn(*args), iters=12, warm=3)
                else:
                    med, mn = time_call(lambda: bev.multi_scale_deformable_attn_int8(qv, sh, ref, qo, qw, s_v, s_o, s_w,
                                                                                     0.02), iters=12, warm=3)
                print(json.dumps({"ablate": mask, "dtype": dt, "us": round(med, 1), "min_us": round(mn, 1)}), flush=True)
            except Exception as exc:  # noqa: BLE001
                print(json.dumps({"ablate": mask, "dtype": dt, "error": str(exc)[:100]}), flush=True)
            finally:
                lib.bevops_msda_set_variant(0)


def kernels(lib):
    """A few launches of each base-SCA flavour for `rocprofv3 --kernel-trace --stats` (per-kernel times)."""
    args, _ = gen(SHAPES["base_sca"], torch.float16, "uniform")
    value, sh, ref, off, logit = gen(SHAPES["base_sca"], torch.float32, "uniform")[0]

    def q(t):
        s = float(t.abs().max()) / 127.0
        return torch.clamp(torch.round(t / s), -127, 127).to(torch.int8), s
    qv, s_v = q(value); qo, s_o = q(off); qw, s_w = q(logit)
    for v in (16, 17):
        lib.bevops_msda_set_variant(v)
        for _ in range(6):
            bev.multi_scale_deformable_attn(*args)
    lib.bevops_msda_set_variant(0)
    for r in (ref, ref.half()):
        for _ in range(6):
            bev.multi_scale_deformable_attn_int8(qv, sh, r, qo, qw, s_v, s_o, s_w, 0.02)
    # int8 DCNv2 at the two ResNet-101 shapes (kernel breakdown of bevops_mdconv_forward_int8)
    g = torch.Generator().manual_seed(0)
    for (B, C, H, W) in ((6, 256, 58, 100), (6, 512, 29, 50)):
        x = torch.randint(-127, 128, (B, C, H, W), generator=g, dtype=torch.int8).cuda()
        off = torch.randint(-127, 128, (B, 18, H, W), generator=g, dtype=torch.int8).cuda()
        mask = torch.randint(0, 128, (B, 9, H, W), generator=g, dtype=torch.int8).cuda()
        w = torch.randint(-127, 128, (C, C, 3, 3), generator=g, dtype=torch.int8).cuda()
        b = torch.zeros(C).cuda()
        for v in (0, 6):     # fused implicit GEMM, im2col + GEMM
            lib.bevops_mdconv_set_variant(v)
            for _ in range(4):
                bev.modulated_deformable_conv2d_int8(x, off, mask, w, b, 0.02, 0.03, 1 / 127, 0.01, 0.05, 1, 1, 1, 1, 1)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(10):
                bev.modulated_deformable_conv2d_int8(x, off, mask, w, b, 0.02, 0.03, 1 / 127, 0.01, 0.05, 1, 1, 1, 1, 1)
            torch.cuda.synchronize()
            print(json.dumps({"op": "dcn_int8", "shape": [B, C, H, W], "variant": v,
                              "us_per_call_wall": round((time.perf_counter() - t0) / 10 * 1e6, 1)}), flush=True)
        lib.bevops_mdconv_set_variant(0)
    torch.cuda.synchronize()


def main():
    lib = load_library()
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        return ablations(lib)
    if len(sys.argv) > 1 and sys.argv[1] == "kernels":
        return kernels(lib)
    plan = [("base_sca", "uniform", [16, 17, 172, 175, 176]), ("base_sca", "rig", [16, 17]),
            ("base_tsa", "uniform", [0, 10, 17]), ("small_tsa", "uniform", [0, 17]), ("small_sca", "uniform", [0, 17])]
    for name, dist, variants in plan:
        args, byt = gen(SHAPES[name], torch.float16, dist)
        for v in variants:
            lib.bevops_msda_set_variant(v)
            try:
                med, mn = time_call(lambda: bev.multi_scale_deformable_attn(*args))
                print(json.dumps({"call": name, "dtype": "f16", "refs": dist, "variant": v, "us": round(med, 1),
                                  "min_us": round(mn, 1), "GBps": round(byt / med / 1e3, 1)}), flush=True)
            except Exception as exc:  # noqa: BLE001
                print(json.dumps({"call": name, "variant": v, "error": str(exc)[:120]}), flush=True)
            finally:
                lib.bevops_msda_set_variant(0)
    for name, dist in (("base_sca", "uniform"), ("base_sca", "rig"), ("base_tsa", "uniform")):
        args, _ = gen(SHAPES[name], torch.float32, dist)
        value, sh, ref, off, logit = args

        def q(t):
            s = float(t.abs().max()) / 127.0
            return torch.clamp(torch.round(t / s), -127, 127).to(torch.int8), s
        qv, s_v = q(value); qo, s_o = q(off); qw, s_w = q(logit)
        for rdt in (torch.float32, torch.float16):
            r = ref.to(rdt)
            byt = qv.numel() + qo.numel() + qw.numel() + r.numel() * r.element_size() + \
                value.shape[0] * off.shape[1] * 8 * 32 + 8 * sh.shape[0]
            for v in (10, 0, 17):
                lib.bevops_msda_set_variant(v)
                try:
                    med, mn = time_call(lambda: bev.multi_scale_deformable_attn_int8(qv, sh, r, qo, qw, s_v, s_o, s_w, 0.02))
                    print(json.dumps({"call": name, "dtype": "i8", "ref": str(rdt)[6:], "refs": dist, "variant": v,
                                      "us": round(med, 1), "min_us": round(mn, 1), "GBps": round(byt / med / 1e3, 1)}),
                          flush=True)
                except Exception as exc:  # noqa: BLE001
                    print(json.dumps({"call": name, "dtype": "i8", "variant": v, "error": str(exc)[:120]}), flush=True)
                finally:
                    lib.bevops_msda_set_variant(0)


if __name__ == "__main__":
    main()
