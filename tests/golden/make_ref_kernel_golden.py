#!/usr/bin/env python3
"""Regenerates tests/golden/refk_*.npz: seeded inputs + the outputs of the REFERENCE's own plugin
kernels (TensorRT/plugin/*/*Kernel.cu) executed on the host through oracle/_ref/libbevref.so
(built by `make -C oracle` from the sources under /root/reference; oracle/cuda_on_cpu/).

Run in the build container (needs /root/reference):  python tests/golden/make_ref_kernel_golden.py
The fixtures travel to the GPU box; the reference does not.
Every case stores fp32 inputs, their int8 quantisation + scales (min-max, like the reference's
test calibrator, det2trt/models/utils/test_trt_ops/utils.py:18-51) and the kernel outputs:
  out_f32   <float> kernel            out_f16 / out_h2   <__half> / <__half2> kernels on fp16-rounded inputs
  out_s8... the int8 kernels
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import refkernels as R  # noqa: E402
from util_bevpool import make_indices  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def q8(x, s=None):
    s = float(np.abs(x).max()) / 127.0 if s is None else s
    return np.clip(np.rint(x / s), -127, 127).astype(np.int8), np.float32(s)


def h(x):
    return x.astype(np.float16)


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, "refk_" + name + ".npz"), **kw)
    print("wrote refk_%s.npz" % name, {k: getattr(v, "shape", v) for k, v in kw.items() if k.startswith("out")})


def msda_case(name, bs, shapes, nq, P, ppg, heads=8, C=32, ref_range=(0.0, 1.0), seed=0):
    rng = np.random.default_rng(seed)
    shapes = np.array(shapes, np.int32)
    nk, L = int((shapes[:, 0] * shapes[:, 1]).sum()), len(shapes)
    value = rng.standard_normal((bs, nk, heads, C)).astype(np.float32)
    ref = rng.uniform(*ref_range, (bs, nq, 1, 2 * ppg)).astype(np.float32)
    off = rng.standard_normal((bs, nq, heads, L * P * 2)).astype(np.float32)
    logit = rng.standard_normal((bs, nq, heads, L * P)).astype(np.float32)
    out_f32 = R.msda(value, shapes, ref, off, logit, R.F32)
    kw = dict(value=value, shapes=shapes, ref=ref, off=off, logit=logit, out_f32=out_f32,
              out_f16=R.msda(h(value), shapes, h(ref), h(off), h(logit), R.F16))
    if C % 2 == 0:
        kw["out_h2"] = R.msda(h(value), shapes, h(ref), h(off), h(logit), R.H2)
    if C % 4 == 0 and P % 4 == 0:
        (vq, sv), (oq, so), (wq, sw) = q8(value), q8(off), q8(logit)
        s_out = np.float32(np.abs(out_f32).max() / 127.0)
        kw.update(value_q=vq, off_q=oq, logit_q=wq, s_value=sv, s_off=so, s_logit=sw, s_out=s_out,
                  out_s8_f32ref=R.msda_s8(vq, sv, shapes, ref, oq, so, wq, sw, s_out, ref_half=False),
                  out_s8_f16ref=R.msda_s8(vq, sv, shapes, h(ref), oq, so, wq, sw, s_out, ref_half=True))
    save("msda_" + name, **kw)


def rotate_case(name, C, H, W, angle, center, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((C, H, W)).astype(np.float32)
    iq, s_in = q8(img)
    kw = dict(img=img, angle=np.float32(angle), center=np.array(center, np.float32), img_q=iq, s_in=s_in)
    for interp, nm in ((0, "bilinear"), (1, "nearest")):
        kw["out_f32_" + nm] = R.rotate(img, angle, center, interp, R.F32)
        kw["out_f16_" + nm] = R.rotate(h(img), angle, center, interp, R.F16)
        kw["out_h2_" + nm] = R.rotate(h(img), angle, center, interp, R.H2)
        kw["out_s8_" + nm] = R.rotate_s8(iq, angle, center, interp, s_in, s_in)
    save("rotate_" + name, **kw)


def grid_sampler_cases():
    rng = np.random.default_rng(0)
    inp = rng.standard_normal((2, 8, 11, 13)).astype(np.float32)
    grid = rng.uniform(-12, 12, (2, 2, 17, 19)).astype(np.float32)
    iq, s_in = q8(inp)
    gq, s_g = q8(grid, 12.0 / 127)
    kw = dict(inp=inp, grid=grid, inp_q=iq, grid_q=gq, s_in=s_in, s_grid=s_g)
    for interp in (0, 1, 2):
        for pad in (0, 1, 2):
            for align in (0, 1):
                tag = "_%d%d%d" % (interp, pad, align)
                kw["out_f32" + tag] = R.grid_sampler(inp, grid, interp, pad, align, R.F32)
                kw["out_f16" + tag] = R.grid_sampler(h(inp), h(grid), interp, pad, align, R.F16)
                kw["out_h2" + tag] = R.grid_sampler(h(inp), h(grid), interp, pad, align, R.H2)
                kw["out_s8" + tag] = R.grid_sampler_s8(iq, gq, interp, pad, align, s_in, s_g, s_in)
    save("grid_sampler_2d", **kw)
    inp3 = rng.standard_normal((2, 3, 5, 6, 7)).astype(np.float32)
    grid3 = rng.uniform(-12, 12, (2, 3, 4, 5, 6)).astype(np.float32)
    kw = dict(inp=inp3, grid=grid3)
    for interp in (0, 1):
        for pad in (0, 1, 2):
            for align in (0, 1):
                tag = "_%d%d%d" % (interp, pad, align)
                kw["out_f32" + tag] = R.grid_sampler(inp3, grid3, interp, pad, align, R.F32)
                kw["out_f16" + tag] = R.grid_sampler(h(inp3), h(grid3), interp, pad, align, R.F16)
    save("grid_sampler_3d", **kw)


def mdconv_case(name, B, Cin, Cout, H, W, K, stride, pad, dil, g, dg, seed=0):
    rng = np.random.default_rng(seed)
    Ho = (H + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    off = (rng.standard_normal((B, dg * 2 * K * K, Ho, Wo)) * 2).astype(np.float32)
    mask = rng.uniform(0, 1, (B, dg * K * K, Ho, Wo)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin // g, K, K)) / np.sqrt(Cin * K * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    out_f32 = R.mdconv(x, off, mask, w, b, stride, pad, dil, g, dg, R.F32)
    kw = dict(x=x, offset=off, mask=mask, weight=w, bias=b, cfg=np.array([stride, pad, dil, g, dg], np.int32),
              out_f32=out_f32, out_f32_nobias=R.mdconv(x, off, mask, w, None, stride, pad, dil, g, dg, R.F32),
              out_f16=R.mdconv(h(x), h(off), h(mask), h(w), h(b), stride, pad, dil, g, dg, R.F16))
    if (Cin // g) % 4 == 0 and (Cin // dg) % 4 == 0:
        (xq, sx), (oq, so), (wq, sw) = q8(x), q8(off), q8(w)
        mq, sm = q8(mask, 1.0 / 127)
        s_out = np.float32(np.abs(out_f32).max() / 127.0)
        kw.update(x_q=xq, offset_q=oq, mask_q=mq, weight_q=wq, s_x=sx, s_offset=so, s_mask=sm, s_weight=sw,
                  s_out=s_out,
                  out_s8=R.mdconv_s8(xq, sx, oq, so, mq, sm, wq, sw, b, s_out, stride, pad, dil, g, dg))
    save("mdconv_" + name, **kw)


def bev_pool_case():
    rng = np.random.default_rng(0)
    N, D, H, W, C, oh, ow = 2, 12, 8, 11, 16, 16, 16
    rd, rf, rb, st, ln = make_indices(N, D, H, W, oh, ow)
    depth = rng.uniform(0, 1, (N, D, H, W)).astype(np.float32)
    feat = rng.standard_normal((N, H, W, C)).astype(np.float32)
    out_f32 = R.bev_pool_v2(depth, feat, rd, rf, rb, st, ln, oh, ow, R.F32)
    dq, sd = q8(depth, 1.0 / 127)
    fq, sf = q8(feat)
    so = np.float32(np.abs(out_f32).max() / 127.0)
    save("bev_pool", depth=depth, feat=feat, ranks_depth=rd, ranks_feat=rf, ranks_bev=rb, interval_starts=st,
         interval_lengths=ln, out_hw=np.array([oh, ow], np.int32), out_f32=out_f32,
         out_f16=R.bev_pool_v2(h(depth), h(feat), rd, rf, rb, st, ln, oh, ow, R.F16),
         out_h2=R.bev_pool_v2(h(depth), h(feat), rd, rf, rb, st, ln, oh, ow, R.H2),
         depth_q=dq, feat_q=fq, s_depth=sd, s_feat=sf, s_out=so,
         out_s8=R.bev_pool_v2(dq, fq, rd, rf, rb, st, ln, oh, ow, scales=(sd, sf, so)))


if __name__ == "__main__":
    msda_case("sca_like", 2, [[12, 20], [6, 10], [3, 5], [2, 3]], 160, 8, 4)
    msda_case("tsa_like", 2, [[20, 20]], 200, 4, 1)
    msda_case("oob", 1, [[7, 9], [4, 5]], 120, 4, 2, ref_range=(-0.3, 1.3))
    msda_case("generic_c12", 2, [[9, 11], [4, 5]], 70, 4, 2, heads=3, C=12)
    msda_case("odd_lp", 1, [[5, 7], [3, 4], [2, 2]], 50, 3, 3, heads=2, C=6)
    rotate_case("small", 8, 30, 40, 13.7, (20.0, 15.0))
    rotate_case("offcenter", 4, 24, 24, -75.2, (17.3, 9.1))
    grid_sampler_cases()
    mdconv_case("plain", 2, 8, 6, 9, 11, 3, 1, 1, 1, 1, 1)
    mdconv_case("grouped_s2", 1, 8, 8, 10, 7, 3, 2, 1, 1, 2, 2)
    mdconv_case("dilated_dg4", 2, 16, 8, 8, 8, 3, 1, 2, 2, 1, 4)
    mdconv_case("k1_g3", 1, 12, 6, 5, 6, 1, 1, 0, 1, 3, 1)
    mdconv_case("c32", 3, 32, 32, 14, 18, 3, 1, 1, 1, 1, 1)
    bev_pool_case()
