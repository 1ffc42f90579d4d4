#!/usr/bin/env python3
"""Per-site attribution of the INT8 engine's error budget (the stand-in for "INT8 NDS/mAP within the reference's
PTQ drop", which needs nuScenes + checkpoints): ONE build of the model bench.py times, ONE calibration, then the
evaluation sequence is replayed with only one GROUP of quantised sites switched to int8 at a time (every other
site on its fp16 operator -- the layer-precision fallback a TensorRT build offers), plus everything-on and the
leave-one-group-out rows.  Reference = the default fp16 model on the same frames; `noise_floor` = the fp32 model
against it (what two correct float evaluations of this random-weight network differ by).
usage: int8_attribution.py [tiny|small|base] [--calib K] [--frames N] [--calibrator entropy|minmax|percentile]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402
from bevformer_tensorrt_amd.quantization import (Int8ChainBackbone, Int8PluginOps, quantize_backbone_convs,  # noqa: E402
                                                 quantize_dense_layers)


def frame(i, H, W, dev, dtype, gen):
    img = torch.randn(1, 6, 3, H, W, generator=gen).to(dev, dtype)
    can = torch.zeros(18)
    can[0], can[1], can[-2], can[-1] = 0.4 * i, -0.15 * i, 0.02 * i, 1.1 * i
    return img, can


def group_of(site, qops):
    """quantised site name -> attribution group"""
    if site.startswith("msda#"):
        return {2: "plugin.tsa_msda", 1: "plugin.dec_msda"}.get(qops.site_batch(site), "plugin.sca_msda")
    if site.startswith("rotate#"):
        return "plugin.rotate"
    if site.startswith("dcn#"):
        return "plugin.dcn"
    name = site.split(":", 1)[1]
    p = name.split(".")
    if p[0] == "encoder":
        return "dense.encoder." + p[2]                      # tsa / sca / ffn
    if p[0] == "decoder":
        return "dense.decoder." + p[2]                      # cross_attn / ffn
    if p[0] == "backbone":
        return "conv.backbone.stage" + p[2]
    if p[0] == "neck":
        return "conv.neck"
    return "dense.other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="base")
    ap.add_argument("--calib", type=int, default=16)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--calibrator", default="entropy")
    ap.add_argument("--no-fp32", action="store_true")
    ap.add_argument("--chain", action="store_true", help="backbone as the int8 activation chain (bench.py's engine)")
    ap.add_argument("--device", default="cuda", help="cpu: plumbing check only (fp32, no site ever switched to int8)")
    a = ap.parse_args()
    name = a.model
    dev = torch.device(a.device)
    dtype = torch.float16 if dev.type == "cuda" else torch.float32
    H, W = B.CONFIGS[name]["image"]
    n_enc = B.CONFIGS[name]["enc_layers"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    fp_ops = None
    if dev.type != "cuda":          # plumbing check: the torch statement of the operators (test infrastructure)
        from oracle.ref_ops import RefOps as fp_ops
    qops = Int8PluginOps(a.calibrator, fp_ops=fp_ops, channels_last=True, fused_sca=True, engine=a.chain)
    model_q = B.BEVFormer(name, ops=qops, seed=0, backbone_layout="nhwc").to(dev, dtype)
    qops.attach(model_q)
    # every dense layer of the encoder / decoder is calibrated (also the ones the engine keeps in fp16), so that
    # each group can be switched on its own
    dense_q = quantize_dense_layers(model_q, qops.cal, lambda n, m: n.startswith(("encoder.", "decoder.")))
    chain, taps = None, []
    if a.chain:       # the int8 activation chain through the backbone, as bench.py's engine runs it (one group)
        chain = Int8ChainBackbone(model_q, qops.cal)
        taps = [m for m in quantize_backbone_convs(model_q, qops.cal, lambda n, m: n.startswith("neck."), conv3x3=True)
                if not hasattr(m, "lin")]          # the FPN's 3x3 convolutions (ConvTapsQ on a pre-quantised input)
        for m in taps:
            m.prequant = True
            m.calibrate()
    else:
        dense_q += quantize_backbone_convs(model_q, qops.cal)
    for m in dense_q:
        m.calibrate()
    run_q = B.FrameRunner(model_q, dev, dtype)
    gen = torch.Generator().manual_seed(1)
    for i in range(a.calib):
        img, can = frame(i, H, W, dev, dtype, gen)
        run_q.step(img, can, l2i, "calib")
    qops.freeze()
    for m in dense_q:
        m.freeze()
    if chain is not None:
        chain.freeze()
    for m in taps:
        m.freeze()
    lin_of = lambda m: getattr(m, "lin", m)          # Conv2dQ keeps its LinearQ in .lin
    dense_sites = {lin_of(m).site: lin_of(m) for m in dense_q}
    plugin_sites = sorted({k.rsplit(".", 1)[0] for k in qops._scales if k.split("#")[0] in ("msda", "rotate", "dcn")})
    groups = {}
    for s in list(dense_sites) + plugin_sites:
        groups.setdefault(group_of(s, qops), []).append(s)

    frames = [frame(100 + i, H, W, dev, dtype, gen) for i in range(a.frames)]

    def evaluate(runner):
        outs = []
        for img, can in frames:
            cls, crd = runner.step(img, can, l2i, "eval")
            outs.append((cls.float().clone(), crd.float().clone(), runner.prev_bev.float().clone()))
        return outs

    ref = evaluate(B.FrameRunner(B.BEVFormer(name, ops=fp_ops, seed=0).to(dev, dtype), dev, dtype))

    def delta(outs):
        rel, cls_err, crd_err, top1 = [], [], [], []
        for (cq, bq, eq), (cf, bf, ef) in zip(outs, ref):
            rel.append(((eq - ef).abs().mean() / ef.std()).item())
            cls_err.append((cq - cf).abs().mean().item())
            crd_err.append((bq - bf).abs().mean().item())
            top1.append((cq[-1].argmax(-1) == cf[-1].argmax(-1)).float().mean().item())
        m = lambda v: round(sum(v) / len(v), 5)
        return dict(bev_embed_rel_err=m(rel), cls_logit_mae=m(cls_err), box_coord_mae=m(crd_err), top1_class_agreement=m(top1))

    if chain is not None:
        groups["chain.backbone"] = ["chain.backbone"]
    if taps:
        groups["conv.neck.fpn3x3"] = ["conv.neck.fpn3x3"]

    def with_sites(on):
        on = set(on)
        for s, lin in dense_sites.items():
            lin.mode = "int8" if s in on else "float"
        if chain is not None:      # off: the fp16 channels-last backbone on the same weights
            chain.ready = "chain.backbone" in on
            for c in chain.convs:
                c.lin.mode = "float"
        for m in taps:
            m.qmode = "int8" if "conv.neck.fpn3x3" in on else "float"
        qops.site_filter = lambda site: site in on
        return delta(evaluate(B.FrameRunner(model_q, dev, dtype)))

    head = dict(model=name, calibrator=a.calibrator, calib_frames=a.calib, eval_frames=a.frames)
    if not a.no_fp32:
        try:
            m32 = B.BEVFormer(name, seed=0).to(dev, torch.float32)
            r32 = B.FrameRunner(m32, dev, torch.float32)
            outs = []
            for img, can in frames:
                cls, crd = r32.step(img.float(), can, l2i, "eval")
                outs.append((cls.float().clone(), crd.float().clone(), r32.prev_bev.float().clone()))
            print(json.dumps(dict(head, row="noise_floor: fp32 model vs fp16 model", **delta(outs))), flush=True)
            del m32, r32
            torch.cuda.empty_cache()
        except Exception as exc:
            print(json.dumps(dict(head, row="noise_floor", error=repr(exc)[:200])), flush=True)
    everything = list(dense_sites) + plugin_sites + (["chain.backbone"] if chain is not None else []) + \
        (["conv.neck.fpn3x3"] if taps else [])
    engine = None
    if chain is not None:     # the subset bench.py's engine switches on (quantization.build_int8_engine)
        from bevformer_tensorrt_amd.quantization import engine_dense_select
        engine = [s for s in dense_sites if engine_dense_select(s.split(":", 1)[1], None)] + ["chain.backbone"] + \
                 (["conv.neck.fpn3x3"] if taps else []) + \
                 [s for s in plugin_sites if s.startswith("msda#") and qops.site_batch(s) != 1]
    if dev.type != "cuda":
        print(json.dumps(dict(head, row="all int8 sites off", groups={g: len(v) for g, v in groups.items()}, **with_sites([]))))
        return
    if engine is not None:
        print(json.dumps(dict(head, row="THE ENGINE (bench.py's int8 build)", sites=len(engine), **with_sites(engine))), flush=True)
    print(json.dumps(dict(head, row="all int8 sites on", sites=len(everything), **with_sites(everything))), flush=True)
    print(json.dumps(dict(head, row="all int8 sites off (the engine's fp16 path)", sites=0, **with_sites([]))), flush=True)
    for g in sorted(groups):
        print(json.dumps(dict(head, row="only " + g, sites=len(groups[g]), **with_sites(groups[g]))), flush=True)
    for g in sorted(groups):
        rest = [s for s in everything if s not in set(groups[g])]
        print(json.dumps(dict(head, row="all but " + g, sites=len(rest), **with_sites(rest))), flush=True)


if __name__ == "__main__":
    main()
