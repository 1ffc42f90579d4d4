"""Index / grid generation on the PRODUCT path: what BEVFormer.forward
evaluates on the device -- G.project_points(..., projection="fma") on the host-made pillar anchors
-- must reproduce the reference's CPU arrays bit for bit at the base size (golden: SHA-256 digests
of the arrays the reference's own point_sampling_trt produced, tests/golden/make_wrapper_golden.py)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import golden
from bevformer_tensorrt_amd import geometry as G

pytestmark = pytest.mark.gpu
PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_base_projection_on_device_is_bit_exact():
    g = golden("geometry_base")
    bh, bw, ih, iw, step = (int(x) for x in g["meta"])
    ref_3d = G.reference_points_3d(bh, bw, PC_RANGE[5] - PC_RANGE[2], 4, device="cpu")
    pillars = G.pillar_points(ref_3d, PC_RANGE).cuda()                 # as BEVFormer._static
    l2i = torch.from_numpy(g["lidar2img"]).cuda()
    cam, mask = G.project_points(pillars, l2i, (ih, iw), projection="fma")
    cam, mask = cam.cpu().numpy(), mask.cpu().numpy()
    assert np.array_equal(cam[:, :, ::step], g["cam_sample"])
    assert np.array_equal(mask[:, ::step], g["mask_sample"])
    assert [_digest(cam), _digest(mask)] == list(g["sha256"][1:])


def test_model_static_geometry_is_the_host_arrays():
    """BEVFormer caches ref_3d / ref_2d / pillars: they must be the CPU values, uploaded."""
    from bevformer_tensorrt_amd import bevformer as B
    model = B.BEVFormer("tiny", seed=0).cuda().half()
    H, W = B.CONFIGS["tiny"]["image"]
    img = torch.zeros(1, 6, 3, H, W, device="cuda", dtype=torch.float16)
    l2i = G.synthetic_lidar2img((H, W)).cuda()
    model(img, torch.zeros(2500, 1, 256, device="cuda", dtype=torch.float16),
          torch.zeros((), device="cuda", dtype=torch.float16), torch.zeros(18, device="cuda"), l2i)
    ref_3d, ref_2d, pillars = model._static
    want = G.reference_points_3d(50, 50, PC_RANGE[5] - PC_RANGE[2], 4, device="cpu")
    assert torch.equal(ref_3d.cpu(), want)
    assert torch.equal(ref_2d.cpu(), G.reference_points_2d(want))
    assert torch.equal(pillars.cpu(), G.pillar_points(want, PC_RANGE))
    g = golden("geometry")
    assert np.array_equal(ref_3d.cpu().numpy(), g["tiny_ref3d"])


def test_frame_runner_shift_is_host_value():
    from bevformer_tensorrt_amd import bevformer as B
    model = B.BEVFormer("tiny", seed=0).cuda().half()
    runner = B.FrameRunner(model, torch.device("cuda"), torch.float16)
    H, W = B.CONFIGS["tiny"]["image"]
    can = torch.zeros(18)
    can[0], can[1], can[-2], can[-1] = 0.8, -0.3, 0.31, 1.7
    l2i = G.synthetic_lidar2img((H, W))
    img = torch.zeros(1, 6, 3, H, W)
    runner.step(img, torch.zeros(18), l2i, "s")
    runner.step(img, can, l2i, "s")
    want = G.bev_shift(can, 50, 50, (102.4 / 50, 102.4 / 50))
    assert torch.equal(runner._in["shift"].cpu(), want)


def test_fused_reference_point_refinement_is_bit_identical_to_the_op_sequence():
    """bevops_refine_reference_points (one launch per decoder layer) against geometry.refine_reference_points (the
    reference's decoder.py:93-103 as seven framework ops) on fp16 tensors: every bit, including reference points at
    and beyond the clamp bounds and regression outputs of both signs and sizes (index generation: SURVEY 8a-6)."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd import geometry as G
    g = torch.Generator().manual_seed(0)
    for scale in (0.3, 3.0, 12.0):
        n = 900 * 7
        tmp = (torch.randn(1, n, 10, generator=g) * scale).half().cuda()
        ref = torch.rand(1, n, 3, generator=g).half()
        ref[0, :40] = torch.tensor([0.0, 1.0, 1e-5, 1 - 1e-3, 6e-8, 0.99951, 0.5, 2e-5] * 15).view(40, 3).half()
        ref = ref.cuda()
        want = G.refine_reference_points(tmp, ref)
        got, xy = bev.refine_reference_points(tmp, ref)
        assert got.shape == want.shape and xy.shape == (1, n, 1, 2)
        assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()
        assert torch.equal(xy, want[..., :2].unsqueeze(2))
    # and the decoder loop using it matches the loop on the op sequence (same model, fused entry hidden)
    from bevformer_tensorrt_amd import bevformer as B
    import bevformer_tensorrt_amd.functions as hip_ops

    class NoRefine:
        def __getattr__(self, name):
            if name == "refine_reference_points":
                raise AttributeError(name)
            return getattr(hip_ops, name)

    dev, dtype = torch.device("cuda"), torch.float16
    m1 = B.BEVFormer("tiny", seed=0).to(dev, dtype)
    m2 = B.BEVFormer("tiny", ops=NoRefine(), seed=0, backbone_layout="nhwc").to(dev, dtype)
    H, W = B.CONFIGS["tiny"]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
    r1, r2 = B.FrameRunner(m1, dev, dtype), B.FrameRunner(m2, dev, dtype)
    can = torch.zeros(18)
    c1, b1 = r1.step(img, can, l2i, "s")
    c2, b2 = r2.step(img, can, l2i, "s")
    assert torch.equal(b1, b2) and torch.equal(c1, c2)
