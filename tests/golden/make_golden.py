#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own Python code.

Run in the build container only (needs /root/reference); the .npz files it writes
are committed so that tests never read /root/reference at run time.

    python tests/golden/make_golden.py

What is executed from the reference tree (loaded by file path, nothing copied):
  * det2trt/models/functions/multi_scale_deformable_attn.py
        _MultiScaleDeformableAttnFunction.forward   (:29-123)
    Its un-vendored dependency `mmcv._ext.ms_deform_attn_forward` (CUDA, absent) is
    substituted by the reference's own pure-torch statement of the same op,
    det2trt/models/utils/trt_ops.py:multi_scale_deformable_attn_pytorch (:4-85) --
    the branch the reference itself takes when `value` is not on a GPU
    (det2trt/models/modules/spatial_cross_attention.py:539-552).
  * det2trt/models/functions/rotate.py        _Rotate.forward          (:12-80)
  * det2trt/models/functions/grid_sampler.py  _GridSampler2D/3D.forward (:19-37,70-88)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("BEVOPS_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_msda():
    trt_ops = _load("det2trt/models/utils/trt_ops.py", "_ref_trt_ops")

    class _Ext:
        """Stands in for mmcv's CUDA `_ext` with the reference's torch-only branch."""

        @staticmethod
        def ms_deform_attn_forward(value, spatial_shapes, level_start_index,
                                   sampling_locations, attention_weights, im2col_step):
            bs, nk, heads, ch = value.shape
            _, nq, _, L, P, _ = sampling_locations.shape
            sizes = [int(h) * int(w) for h, w in spatial_shapes.tolist()]
            value_list = list(value.split(sizes, dim=1))
            return trt_ops.multi_scale_deformable_attn_pytorch(
                value_list, spatial_shapes, sampling_locations, attention_weights,
                heads, heads * ch, L, P, nq, bs)

        ms_deform_attn_backward = None

    mmcv = types.ModuleType("mmcv")
    mmcv_utils = types.ModuleType("mmcv.utils")
    ext_loader = types.ModuleType("mmcv.utils.ext_loader")
    ext_loader.load_ext = lambda name, funcs: _Ext
    mmcv_utils.ext_loader = ext_loader
    mmcv.utils = mmcv_utils
    saved = {k: sys.modules.get(k) for k in ("mmcv", "mmcv.utils", "mmcv.utils.ext_loader")}
    sys.modules.update({"mmcv": mmcv, "mmcv.utils": mmcv_utils,
                        "mmcv.utils.ext_loader": ext_loader})
    try:
        mod = _load("det2trt/models/functions/multi_scale_deformable_attn.py", "_ref_msda")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


MSDA_CASES = {
    # name: (bs, shapes, nq, heads, C, P, ppg, ref_lo, ref_hi, off_std)
    "sca_like": (2, [[12, 20], [6, 10], [3, 5], [2, 3]], 48, 8, 32, 8, 4, 0.0, 1.0, 1.0),
    "tsa_like": (2, [[10, 10]], 96, 8, 32, 4, 1, 0.0, 1.0, 1.0),
    "tiny_sca": (6, [[8, 12]], 40, 8, 32, 8, 4, 0.0, 1.0, 1.0),
    "edges": (2, [[7, 9], [3, 4]], 64, 2, 8, 4, 2, -0.4, 1.4, 3.0),
    "odd_lp": (1, [[5, 6], [3, 3], [2, 2]], 33, 3, 12, 2, 1, -0.1, 1.1, 1.5),
}


def msda_inputs(case, seed=0):
    bs, shapes, nq, heads, C, P, ppg, lo, hi, off_std = MSDA_CASES[case]
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    nk = sum(h * w for h, w in shapes)
    value = torch.randn(bs, nk, heads, C, generator=g)
    ref = torch.rand(bs, nq, 1, 2 * ppg, generator=g) * (hi - lo) + lo
    off = torch.randn(bs, nq, heads, L * P * 2, generator=g) * off_std
    logit = torch.randn(bs, nq, heads, L * P, generator=g)
    return value, torch.tensor(shapes, dtype=torch.int64), ref, off, logit


def make_msda():
    ref_mod = load_reference_msda()
    fn = ref_mod._MultiScaleDeformableAttnFunction.apply
    for case in MSDA_CASES:
        value, shapes, ref, off, logit = msda_inputs(case)
        out32 = fn(value, shapes, ref, off, logit)
        # reference fp16 eager path: pre-processing in half, sampling in fp32 (:96-101,123)
        out16 = fn(value.half(), shapes, ref.half(), off.half(), logit.half())
        np.savez_compressed(
            os.path.join(OUT, f"msda_{case}.npz"),
            value=value.numpy(), shapes=shapes.numpy().astype(np.int32), ref=ref.numpy(),
            off=off.numpy(), logit=logit.numpy(), out_fp32=out32.numpy(),
            out_fp16_eager=out16.numpy())
        print("msda", case, tuple(out32.shape), float(out32.abs().mean()))


ROTATE_CASES = {
    # name: (C, H, W, angle_deg, center)
    "sq_small": (4, 16, 16, 33.0, (8.0, 8.0)),
    "rect": (3, 12, 20, -117.5, (9.0, 5.0)),
    "offcenter": (2, 10, 10, 200.0, (25.0, 25.0)),   # reference test uses a far-off centre
    "bev_like": (8, 50, 50, 1.7, (25.0, 25.0)),
}


def make_rotate():
    mod = _load("det2trt/models/functions/rotate.py", "_ref_rotate")
    for case, (C, H, W, ang, ctr) in ROTATE_CASES.items():
        g = torch.Generator().manual_seed(0)
        img = torch.randn(C, H, W, generator=g)
        angle = torch.tensor(ang)
        center = torch.tensor(ctr)
        outs = {}
        for name, mode in (("bilinear", 0), ("nearest", 1)):
            outs[name] = mod._Rotate.forward(None, img, angle, center, mode).numpy()
        np.savez_compressed(os.path.join(OUT, f"rotate_{case}.npz"), img=img.numpy(),
                            angle=np.float32(ang), center=np.array(ctr, np.float32), **outs)
        print("rotate", case, outs["bilinear"].shape)


def make_grid_sampler():
    mod = _load("det2trt/models/functions/grid_sampler.py", "_ref_grid")
    g = torch.Generator().manual_seed(0)
    # 2-D: grid spans [-15, 15] like the reference test (50% out of range)
    inp = torch.randn(2, 3, 9, 11, generator=g)
    lin_h = torch.linspace(-15, 15, 23)
    lin_w = torch.linspace(-15, 15, 29)
    gy, gx = torch.meshgrid(lin_h, lin_w, indexing="ij")
    grid = torch.stack([gx, gy], 0)[None].repeat(2, 1, 1, 1)
    grid = grid + torch.randn(grid.shape, generator=g) * 0.7
    res = {"input": inp.numpy(), "grid": grid.numpy()}
    for mi, mname in enumerate(("bilinear", "nearest", "bicubic")):
        for pi, pname in enumerate(("zeros", "border", "reflection")):
            for align in (False, True):
                out = mod._GridSampler2D.apply(inp, grid, mi, pi, align)
                res[f"{mname}_{pname}_{int(align)}"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "grid_sampler_2d.npz"), **res)
    print("grid_sampler 2d", len(res) - 2, "combos")
    # 3-D
    inp = torch.randn(2, 3, 5, 6, 7, generator=g)
    lin = [torch.linspace(-14, 14, n) for n in (6, 7, 9)]
    gz, gy, gx = torch.meshgrid(*lin, indexing="ij")
    grid = torch.stack([gx, gy, gz], 0)[None].repeat(2, 1, 1, 1, 1)
    grid = grid + torch.randn(grid.shape, generator=g) * 0.7
    res = {"input": inp.numpy(), "grid": grid.numpy()}
    for mi, mname in enumerate(("bilinear", "nearest")):
        for pi, pname in enumerate(("zeros", "border", "reflection")):
            for align in (False, True):
                out = mod._GridSampler3D.apply(inp, grid, mi, pi, align)
                res[f"{mname}_{pname}_{int(align)}"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "grid_sampler_3d.npz"), **res)
    print("grid_sampler 3d", len(res) - 2, "combos")


if __name__ == "__main__":
    torch.set_num_threads(4)
    make_msda()
    make_rotate()
    make_grid_sampler()
    make_bev_pool()
    make_geometry()


def make_bev_pool():
    """Run the reference test's own input generator `bev_pool_prepare()` (hard-coded
    NuScenes calibration, det2trt/models/utils/test_trt_ops/test_bev_pool_v2.py:16-...) on
    the CPU and keep the index tensors it produces + the oracle-independent expected
    output computed by torch.index_add_ (semantics of third_party/bev_mmdet3d/ops/
    bev_pool_v2/src/bev_pool_cuda.cu:22-46)."""
    src = open(os.path.join(REF, "det2trt/models/utils/test_trt_ops/test_bev_pool_v2.py")).read()
    src = src.split("class ")[0]                      # generator functions only
    src = src.replace("from .base_test_case import BaseTestCase", "")
    src = src.replace('device="cuda"', 'device="cpu"').replace(".cuda()", "")
    ns = {}
    exec(compile(src, "ref_test_bev_pool_v2", "exec"), ns)
    out = ns["bev_pool_prepare"]()
    names = ["ranks_bev", "ranks_depth", "ranks_feat", "interval_starts", "interval_lengths"]  # :247
    arrs = {n: t.cpu().numpy().astype(np.int32) for n, t in zip(names, out)}
    for n, a in arrs.items():
        print("bev_pool", n, a.shape, a.min(), a.max())
    np.savez_compressed(os.path.join(OUT, "bev_pool_ref_ranks.npz"), **arrs)


def _extract_functions(path, names, min_line=0):
    """exec only the named (static)methods of a reference module whose top-level imports
    (mmcv, ...) cannot be satisfied here.  `min_line` selects the class (the *TRTP wrappers
    come after the *TRT ones in the file)."""
    import ast
    import textwrap
    src = open(os.path.join(REF, path)).read()
    ns = {"torch": torch, "np": np}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.lineno >= min_line:
            seg = textwrap.dedent(ast.get_source_segment(src, node))
            print("  ref function", node.name, "at", path, node.lineno)
            exec(compile(seg, path, "exec"), ns)
    return ns


def make_geometry():
    """Reference index/grid generation executed on CPU: encoder.py:170-259 (static /
    self-light methods) and the shift arithmetic of transformer.py:262-294."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from bevformer_tensorrt_amd.geometry import synthetic_lidar2img
    enc = _extract_functions("det2trt/models/modules/encoder.py",
                             ["get_reference_points_3d", "point_sampling_trt"], min_line=165)
    pc_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]

    class _Self:
        num_points_in_pillar = 4

    res = {}
    for tag, (bh, bw, img) in {"tiny": (50, 50, (480, 800)), "small40": (40, 40, (736, 1280))}.items():
        ref_3d = enc["get_reference_points_3d"](bh, bw, pc_range[5] - pc_range[2], 4, bs=1,
                                                device="cpu", dtype=torch.float)
        l2i = synthetic_lidar2img(img)
        cam, mask = enc["point_sampling_trt"](_Self(), ref_3d, pc_range, l2i, img)
        res[f"{tag}_ref3d"] = ref_3d.numpy()
        res[f"{tag}_lidar2img"] = l2i.numpy()
        res[f"{tag}_cam"] = cam.numpy()
        res[f"{tag}_mask"] = mask.numpy()
        res[f"{tag}_meta"] = np.array([bh, bw, img[0], img[1]], np.int32)
    # shift arithmetic: lift the statements of get_bev_features_trt (transformer.py:262-294)
    src = open(os.path.join(REF, "det2trt/models/modules/transformer.py")).read().split("\n")
    start = next(i for i, l in enumerate(src) if "delta_x = can_bus[0:1]" in l and i > 250)
    end = next(i for i, l in enumerate(src) if "shift = torch.cat([shift_x, shift_y])" in l and i > start)
    import textwrap
    body = textwrap.dedent("\n".join(src[start:end + 1]))
    can = torch.tensor([[0.8, -0.3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.31, 1.7],
                        [-1.2, 0.05, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -2.9, -0.4],
                        [0.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0]])
    shifts = []
    for c in can:
        class _S:
            use_shift = True
        ns = {"torch": torch, "np": np, "can_bus": c, "grid_length": [0.512, 0.512], "bev_h": 200,
              "bev_w": 200, "self": _S()}
        exec(body, ns)
        shifts.append(ns["shift"].numpy())
    res["can_bus"] = can.numpy()
    res["shift"] = np.stack(shifts)
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **res)
    print("geometry", {k: v.shape for k, v in res.items()})
