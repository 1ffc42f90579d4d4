#!/usr/bin/env python3
"""Model-level INT8 error budget (the stand-in for "INT8 NDS/mAP within the reference's PTQ drop",
which needs nuScenes + checkpoints): the re-hosted BEVFormer with its plugin call sites (MSDA,
rotate, DCNv2) on the INT8 operators -- scales from the native PTQ calibrators over K calibration
frames -- against the same weights on the fp16 operators, on frames NOT used for calibration.
With --dense the dense layers of the encoder / decoder blocks (value_proj, sampling_offsets,
attention_weights, output_proj, FFN) additionally run as LinearQ (int8 x int8 GEMM, per-tensor scales from
the same calibrator; det2trt/models/utils/register.py:78-84), reported next to the plugin-only figure.
With --engine the INT8 build is the one bench.py times (`ModelFrames("int8")`): channels-last backbone with its 1x1 /
plain 3x3 convolutions as Conv2dQ / ConvTapsQ, encoder / decoder dense layers as LinearQ, TSA / decoder MSDA and
rotate on the INT8 plugins, fused fp16 SCA sampling -- against the default fp16 model.
usage: int8_model_delta.py [tiny|small|base ...] [--calib K] [--frames N] [--calibrator entropy|minmax|percentile] [--dense|--engine]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402
from bevformer_tensorrt_amd.quantization import Int8PluginOps, quantize_backbone_convs, quantize_dense_layers  # noqa: E402


def frame(i, H, W, dev, dtype, gen):
    img = torch.randn(1, 6, 3, H, W, generator=gen).to(dev, dtype)
    can = torch.zeros(18)
    can[0], can[1], can[-2], can[-1] = 0.4 * i, -0.15 * i, 0.02 * i, 1.1 * i
    return img, can


def run(name, calib, frames, calibrator, dense=False, engine=False):
    dev, dtype = torch.device("cuda"), torch.float16
    H, W = B.CONFIGS[name]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    if engine:
        qops = Int8PluginOps(calibrator, channels_last=True, fused_sca=True)
        model_q = B.BEVFormer(name, ops=qops, seed=0, backbone_layout="nhwc").to(dev, dtype)
    else:
        qops = Int8PluginOps(calibrator)
        model_q = B.BEVFormer(name, ops=qops, seed=0).to(dev, dtype)
    qops.attach(model_q)
    dense_q = []
    if engine:
        dense_q = quantize_dense_layers(model_q, qops.cal, lambda n, m: n.startswith(("encoder.", "decoder.")))
        dense_q += quantize_backbone_convs(model_q, qops.cal)
        for m in dense_q:
            m.calibrate()
    elif dense:   # encoder / decoder blocks only (the backbone's 1x1 convolutions stay in fp16 here)
        dense_q = quantize_dense_layers(model_q, qops.cal, lambda n, m: n.startswith(("encoder.", "decoder.")))
        for m in dense_q:
            m.calibrate()
    model_f = B.BEVFormer(name, seed=0).to(dev, dtype)                 # fp16 operators (fused paths on)
    run_q, run_f = B.FrameRunner(model_q, dev, dtype), B.FrameRunner(model_f, dev, dtype)
    gen = torch.Generator().manual_seed(1)
    for i in range(calib):                       # calibration frames (fp operators, statistics collected)
        img, can = frame(i, H, W, dev, dtype, gen)
        run_q.step(img, can, l2i, "calib")
    scales = qops.freeze()
    for m in dense_q:
        m.freeze()
    run_q = B.FrameRunner(model_q, dev, dtype)   # fresh temporal state for the evaluation sequence
    rel, cls_err, crd_err, top1 = [], [], [], []
    for i in range(frames):
        img, can = frame(100 + i, H, W, dev, dtype, gen)
        cq, bq = run_q.step(img, can, l2i, "eval")
        cf, bf = run_f.step(img, can, l2i, "eval")
        eq, ef = run_q.prev_bev.float(), run_f.prev_bev.float()
        rel.append(((eq - ef).abs().mean() / ef.std()).item())
        cls_err.append((cq.float() - cf.float()).abs().mean().item())
        crd_err.append((bq.float() - bf.float()).abs().mean().item())
        top1.append((cq[-1].argmax(-1) == cf[-1].argmax(-1)).float().mean().item())
    m = lambda v: round(sum(v) / len(v), 5)
    return dict(model=name, build="bench.py INT8 engine" if engine else ("plugins + dense" if dense else "plugins"),
                calibrator=calibrator, calib_frames=calib, eval_frames=frames, int8_sites=len(scales),
                int8_dense_layers=len(dense_q),
                bev_embed_rel_err=m(rel), bev_embed_rel_err_last=round(rel[-1], 5), cls_logit_mae=m(cls_err),
                box_coord_mae=m(crd_err), top1_class_agreement=m(top1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("models", nargs="*", default=["tiny", "small"])
    ap.add_argument("--calib", type=int, default=3)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--calibrator", default="entropy")
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--engine", action="store_true")
    a = ap.parse_args()
    for mname in a.models:
        if a.engine:
            print(json.dumps(run(mname, a.calib, a.frames, a.calibrator, engine=True)), flush=True)
            continue
        print(json.dumps(run(mname, a.calib, a.frames, a.calibrator)), flush=True)
        if a.dense:
            print(json.dumps(run(mname, a.calib, a.frames, a.calibrator, dense=True)), flush=True)
