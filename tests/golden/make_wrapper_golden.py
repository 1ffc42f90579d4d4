#!/usr/bin/env python3
"""Golden vectors for the L3 callers, produced by EXECUTING the
reference's own `forward_trt` methods.  Build container only (needs /root/reference):

    python tests/golden/make_wrapper_golden.py

The reference package cannot be imported (mmcv, pytorch_quantization and a CUDA .so are absent),
so the methods are lifted out of their files by AST -- decorators dropped, nothing else touched --
and run against a stub `self` that carries exactly the attributes the method reads:
  * SpatialCrossAttentionTRTP.forward_trt        det2trt/models/modules/spatial_cross_attention.py:200-273
  * MSDeformableAttention3DTRTP.forward_trt      ...spatial_cross_attention.py:694-768
  * TemporalSelfAttentionTRTP.forward_trt        det2trt/models/modules/temporal_self_attention.py:350-457
  * CustomMSDeformableAttentionTRTP.forward_trt  det2trt/models/modules/decoder.py:381-471
  * DetectionTransformerDecoderTRTP.forward      det2trt/models/modules/decoder.py:52-112 (+ inverse_sigmoid :24-40)
`self.multi_scale_deformable_attn` is the reference's _MultiScaleDeformableAttnFunction.apply
(functions/multi_scale_deformable_attn.py:29-123) with mmcv's absent CUDA extension replaced by the
reference's own torch statement of it (make_golden.load_reference_msda).  Dense members are
torch.nn.Linear with seeded weights, which travel in the fixture.
Also written here: the base-size (200x200 BEV, 928x1600 images) geometry golden as SHA-256 digests
of the arrays' bytes plus a strided sample (the arrays themselves are 11 MB).
"""
import ast
import hashlib
import os
import sys
import textwrap

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402

REF = MG.REF
EMBED, HEADS = 256, 8


def lift(path, cls, name, extra_ns=None):
    """Source of method `name` of class `cls`, decorators removed, compiled into a function."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == name:
                    fn.decorator_list = []
                    seg = ast.unparse(fn)
                    print(f"  lifted {cls}.{name} at {path}:{fn.lineno}-{fn.end_lineno}")
                    ns = {"torch": torch, "np": np, "F": torch.nn.functional}
                    ns.update(extra_ns or {})
                    exec(compile(textwrap.dedent(seg), path, "exec"), ns)
                    return ns[name]
    raise KeyError((path, cls, name))


def lift_function(path, name):
    src = open(os.path.join(REF, path)).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            ns = {"torch": torch}
            exec(compile(ast.unparse(node), path, "exec"), ns)
            print(f"  lifted {name} at {path}:{node.lineno}-{node.end_lineno}")
            return ns[name]
    raise KeyError((path, name))


class Stub:
    """attribute bag standing in for the mmcv module instance"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def linear(i, o, gen, w_scale=1.0, b_scale=0.0):
    m = nn.Linear(i, o)
    with torch.no_grad():
        m.weight.copy_(torch.randn(o, i, generator=gen) * (w_scale / i ** 0.5))
        m.bias.copy_(torch.randn(o, generator=gen) * b_scale)
    return m


def params(prefix, **mods):
    out = {}
    for name, m in mods.items():
        out[f"{prefix}{name}.weight"] = m.weight.detach().numpy()
        out[f"{prefix}{name}.bias"] = m.bias.detach().numpy()
    return out


def make_wrappers():
    msda_fn = MG.load_reference_msda()._MultiScaleDeformableAttnFunction.apply
    g = torch.Generator().manual_seed(0)
    res = {}
    with torch.no_grad():
        # ---------------- SCA: 6 cameras, 2 levels, 8 points, 4 anchors per pillar
        levels = [[6, 10], [3, 5]]
        nk = sum(h * w for h, w in levels)
        nq, cams, D, P = 50, 6, 4, 8
        sca_path = "det2trt/models/modules/spatial_cross_attention.py"
        inner = Stub(
            batch_first=True, num_heads=HEADS, embed_dims=EMBED, num_levels=len(levels), num_points=P,
            value_proj=linear(EMBED, EMBED, g), output_proj=None,
            sampling_offsets=linear(EMBED, HEADS * len(levels) * P * 2, g, 1.0, 2.0),
            attention_weights=linear(EMBED, HEADS * len(levels) * P, g, 2.0, 0.5),
            multi_scale_deformable_attn=msda_fn)
        inner.forward_trt = lift(sca_path, "MSDeformableAttention3DTRTP", "forward_trt").__get__(inner)
        outer = Stub(num_cams=cams, embed_dims=EMBED, deformable_attention=inner,
                     output_proj=linear(EMBED, EMBED, g, 1.0, 0.1), dropout=nn.Identity())
        sca = lift(sca_path, "SpatialCrossAttentionTRTP", "forward_trt").__get__(outer)
        query = torch.randn(1, nq, EMBED, generator=g)
        value = torch.randn(cams, nk, EMBED, generator=g)
        ref_cam = torch.rand(cams, 1, nq, D, 2, generator=g) * 1.4 - 0.2       # some anchors out of view
        vis = (torch.rand(cams, nq, 1, generator=g) < 0.4).float()
        bev_mask = vis / vis.sum(0, keepdim=True).clamp(min=1e-4)               # as encoder.py:255-258
        shapes = torch.tensor(levels, dtype=torch.int64)
        out = sca(query, None, value, reference_points_cam=ref_cam, bev_mask=bev_mask, spatial_shapes=shapes,
                  level_start_index=None)
        res.update(params("sca.", value_proj=inner.value_proj, sampling_offsets=inner.sampling_offsets,
                          attention_weights=inner.attention_weights, output_proj=outer.output_proj))
        res.update({"sca.query": query.numpy(), "sca.value": value.numpy(), "sca.ref_cam": ref_cam.numpy(),
                    "sca.bev_mask": bev_mask.numpy(), "sca.shapes": shapes.numpy().astype(np.int32),
                    "sca.out": out.numpy()})
        print("SCA", tuple(out.shape), float(out.abs().mean()))

        # ---------------- TSA: bev queue 2, one level, 4 points
        bh, bw, P = 6, 7, 4
        nq = bh * bw
        tsa_self = Stub(
            batch_first=True, num_heads=HEADS, embed_dims=EMBED, num_levels=1, num_points=P, num_bev_queue=2,
            value_proj=linear(EMBED, EMBED, g), output_proj=linear(EMBED, EMBED, g, 1.0, 0.1),
            sampling_offsets=linear(2 * EMBED, 2 * HEADS * P * 2, g, 1.0, 1.5),
            attention_weights=linear(2 * EMBED, 2 * HEADS * P, g, 2.0, 0.5), dropout=nn.Identity(),
            multi_scale_deformable_attn=msda_fn)
        tsa = lift("det2trt/models/modules/temporal_self_attention.py", "TemporalSelfAttentionTRTP",
                   "forward_trt").__get__(tsa_self)
        query = torch.randn(1, nq, EMBED, generator=g)
        prev = torch.randn(2, nq, EMBED, generator=g)
        pos = torch.randn(1, nq, EMBED, generator=g)
        ref_2d = torch.rand(2, nq, 1, 2, generator=g)
        shapes = torch.tensor([[bh, bw]], dtype=torch.int64)
        out = tsa(query, prev, prev, None, query_pos=pos, reference_points=ref_2d, spatial_shapes=shapes,
                  level_start_index=torch.tensor([0]))
        res.update(params("tsa.", value_proj=tsa_self.value_proj, sampling_offsets=tsa_self.sampling_offsets,
                          attention_weights=tsa_self.attention_weights, output_proj=tsa_self.output_proj))
        res.update({"tsa.query": query.numpy(), "tsa.prev": prev.numpy(), "tsa.pos": pos.numpy(),
                    "tsa.ref_2d": ref_2d.numpy(), "tsa.shapes": shapes.numpy().astype(np.int32),
                    "tsa.out": out.numpy()})
        print("TSA", tuple(out.shape), float(out.abs().mean()))

        # ---------------- decoder cross attention: (num_query, 1, C) layout, one level, 4 points
        ndec = 30
        dec_self = Stub(
            batch_first=False, num_heads=HEADS, embed_dims=EMBED, num_levels=1, num_points=P,
            value_proj=linear(EMBED, EMBED, g), output_proj=linear(EMBED, EMBED, g, 1.0, 0.1),
            sampling_offsets=linear(EMBED, HEADS * P * 2, g, 1.0, 1.5),
            attention_weights=linear(EMBED, HEADS * P, g, 2.0, 0.5), dropout=nn.Identity(),
            multi_scale_deformable_attn=msda_fn)
        dec = lift("det2trt/models/modules/decoder.py", "CustomMSDeformableAttentionTRTP",
                   "forward_trt").__get__(dec_self)
        query = torch.randn(ndec, 1, EMBED, generator=g)
        qpos = torch.randn(ndec, 1, EMBED, generator=g)
        bev = torch.randn(nq, 1, EMBED, generator=g)
        refp = torch.rand(1, ndec, 1, 2, generator=g)
        out = dec(query, None, bev, None, query_pos=qpos, reference_points=refp, spatial_shapes=shapes,
                  level_start_index=torch.tensor([0]))
        res.update(params("dec.", value_proj=dec_self.value_proj, sampling_offsets=dec_self.sampling_offsets,
                          attention_weights=dec_self.attention_weights, output_proj=dec_self.output_proj))
        res.update({"dec.query": query.numpy(), "dec.query_pos": qpos.numpy(), "dec.bev": bev.numpy(),
                    "dec.ref": refp.numpy(), "dec.shapes": shapes.numpy().astype(np.int32), "dec.out": out.numpy()})
        print("decoder attention", tuple(out.shape), float(out.abs().mean()))

        # ---------------- decoder loop: reference-point refinement over 3 layers (decoder.py:52-112)
        inv = lift_function("det2trt/models/modules/decoder.py", "inverse_sigmoid")
        fwd = lift("det2trt/models/modules/decoder.py", "DetectionTransformerDecoderTRTP", "forward",
                   {"inverse_sigmoid": inv})
        steps = [torch.randn(ndec, 1, EMBED, generator=g) * 0.5 for _ in range(3)]

        class _Layer:                 # stands in for the decoder layer: out = query + a fixed step
            def __init__(self, d):
                self.d = d
                self.seen = []

            def __call__(self, q, *a, reference_points=None, **kw):
                self.seen.append(reference_points.clone())
                return q + self.d

        layers = [_Layer(d) for d in steps]
        regs = [linear(EMBED, 10, g, 3.0, 1.0) for _ in range(3)]
        loop_self = Stub(layers=layers, return_intermediate=True)
        ref0 = torch.rand(1, ndec, 3, generator=g)
        ref0[0, 0] = torch.tensor([0.0, 1.0, 2e-6])         # below / above the clamp of decoder.py:37
        ref0[0, 1] = torch.tensor([1.0 - 2e-6, 5e-6, 1.0])
        query = torch.randn(ndec, 1, EMBED, generator=g)
        inter, inter_ref = fwd(loop_self, query, reference_points=ref0, reg_branches=regs)
        for i, r in enumerate(regs):
            res[f"loop.reg{i}.weight"] = r.weight.detach().numpy()
            res[f"loop.reg{i}.bias"] = r.bias.detach().numpy()
        res.update({"loop.query": query.numpy(), "loop.steps": torch.stack(steps).numpy(), "loop.ref0": ref0.numpy(),
                    "loop.inter": inter.numpy(), "loop.inter_ref": inter_ref.numpy(),
                    "loop.layer_ref_in": torch.stack([l.seen[0] for l in layers]).numpy()})
        print("decoder loop", tuple(inter_ref.shape))
    np.savez_compressed(os.path.join(HERE, "wrappers.npz"), **res)


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def make_geometry_base():
    """encoder.py:170-259 executed at the base config (bev 200x200, 928x1600 images, 6 cameras)."""
    from bevformer_tensorrt_amd.geometry import synthetic_lidar2img
    enc = MG._extract_functions("det2trt/models/modules/encoder.py",
                                ["get_reference_points_3d", "point_sampling_trt"], min_line=165)
    pc_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]

    class _Self:
        num_points_in_pillar = 4

    bh, bw, img = 200, 200, (928, 1600)
    ref_3d = enc["get_reference_points_3d"](bh, bw, pc_range[5] - pc_range[2], 4, bs=1, device="cpu",
                                            dtype=torch.float)
    l2i = synthetic_lidar2img(img)
    cam, mask = enc["point_sampling_trt"](_Self(), ref_3d, pc_range, l2i, img)
    ref_3d, cam, mask = ref_3d.numpy(), cam.numpy(), mask.numpy()
    step = 37
    res = {"meta": np.array([bh, bw, img[0], img[1], step], np.int32), "lidar2img": l2i.numpy(),
           "sha256": np.array([digest(ref_3d), digest(cam), digest(mask)]),
           "ref3d_sample": ref_3d[:, :, ::step], "cam_sample": cam[:, :, ::step], "mask_sample": mask[:, ::step]}
    np.savez_compressed(os.path.join(HERE, "geometry_base.npz"), **res)
    print("geometry base", cam.shape, mask.shape, res["sha256"])


def reference_test_calibration():
    """The 6-camera nuScenes calibration hard-coded in the reference's bev_pool test
    (det2trt/models/utils/test_trt_ops/test_bev_pool_v2.py:56-...): run its `bev_pool_prepare()` with
    `get_lidar_coor` replaced by a recorder and keep the five tensors it is called with."""
    src = open(os.path.join(REF, "det2trt/models/utils/test_trt_ops/test_bev_pool_v2.py")).read()
    src = src.split("class ")[0].replace("from .base_test_case import BaseTestCase", "")
    src = src.replace('device="cuda"', 'device="cpu"').replace(".cuda()", "")
    ns = {}
    exec(compile(src, "ref_test_bev_pool_v2", "exec"), ns)

    class _Got(Exception):
        pass

    def recorder(*args):
        raise _Got(args)

    ns["get_lidar_coor"] = recorder
    try:
        ns["bev_pool_prepare"]()
    except _Got as g:
        return [t.clone() for t in g.args[0]]
    raise RuntimeError("bev_pool_prepare did not call get_lidar_coor")


def make_bevdet_geometry():
    """LSSViewTransformer's own geometry methods (third_party/bev_mmdet3d/models/necks/
    view_transformer.py:66-168,239-312) lifted and run on CPU at the BEVDet-R50 config
    (configs/bevdet/bevdet-r50-cbgs.py:44-104) with the reference test's calibration."""
    path = "third_party/bev_mmdet3d/models/necks/view_transformer.py"
    names = ["create_grid_infos", "create_frustum", "get_lidar_coor", "voxel_pooling_prepare_v2"]
    fns = {n: lift(path, "LSSViewTransformer", n) for n in names}
    cfg = dict(x=[-51.2, 51.2, 0.8], y=[-51.2, 51.2, 0.8], z=[-5, 3, 8], depth=[1.0, 60.0, 1.0])
    me = Stub(sid=False)
    fns["create_grid_infos"](me, **cfg)
    me.frustum = fns["create_frustum"](me, cfg["depth"], (256, 704), 16)
    sensor2ego, cam2imgs, post_rots, post_trans, bda = reference_test_calibration()
    # the test's rig is calibrated for 512x1408 inputs; BEVDet-R50 runs the same cameras at 256x704:
    # the image-view augmentation (post_rots / post_trans) scales by one half
    post_rots = post_rots.clone()
    post_rots[..., :2, :2] *= 0.5
    post_trans = post_trans.clone() * 0.5
    coor = fns["get_lidar_coor"](me, sensor2ego, None, cam2imgs, post_rots, post_trans, bda)
    ranks = fns["voxel_pooling_prepare_v2"](me, coor)
    names5 = ["ranks_bev", "ranks_depth", "ranks_feat", "interval_starts", "interval_lengths"]
    res = {"sensor2ego": sensor2ego.numpy(), "cam2imgs": cam2imgs.numpy(), "post_rots": post_rots.numpy(),
           "post_trans": post_trans.numpy(), "bda": bda.numpy(), "D": np.int32(me.D),
           "frustum_sha256": np.array(digest(me.frustum.contiguous().numpy())),
           "coor_sha256": np.array(digest(coor.contiguous().numpy())),
           "coor_sample": coor[0, :, ::7, ::3, ::5].contiguous().numpy()}
    for n, t in zip(names5, ranks):
        a = t.numpy().astype(np.int32)
        res[n + "_sha256"] = np.array(digest(a))
        res[n + "_len"] = np.int64(a.size)
        res[n + "_head"] = a[:64]
    np.savez_compressed(os.path.join(HERE, "bevdet_geometry.npz"), **res)
    print("bevdet geometry: points", int(res["ranks_bev_len"]), "intervals", int(res["interval_starts_len"]), "D", me.D)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "bevdet":
        make_bevdet_geometry()
        raise SystemExit(0)
    make_wrappers()
    make_geometry_base()
    make_bevdet_geometry()
