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
