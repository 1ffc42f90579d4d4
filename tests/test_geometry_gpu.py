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


def _same_bits(a, b):
    """Equal bit for bit (NaN payloads and the sign of zero included)."""
    view = torch.int32 if a.dtype == torch.float32 else torch.int16
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.contiguous().view(view), b.contiguous().view(view))


def _rigs(ih, iw):
    """Calibration sets: the synthetic ring rig, the golden nuScenes-like matrices, and perturbed copies (what a frame
    loop sees: other matrices every frame) -- incl. pillars behind cameras and on the image border."""
    g = torch.Generator().manual_seed(5)
    base = G.synthetic_lidar2img((ih, iw))
    rigs = [base, torch.from_numpy(golden("geometry_base")["lidar2img"]).reshape(1, 6, 4, 4)]
    for k in range(3):
        rigs.append(base * (1 + 0.02 * torch.randn(1, 6, 4, 4, generator=g)) + 0.05 * torch.randn(1, 6, 4, 4, generator=g))
    return rigs


@pytest.mark.parametrize("bev,image", [((200, 200), (928, 1600)), ((150, 150), (736, 1280)), ((50, 50), (480, 800)),
                                       ((37, 21), (480, 800))])
def test_point_sampling_kernel_is_bit_identical_to_the_torch_op_sequence(bev, image):
    """bevops_point_sampling (one launch, the first node of the frame's graph since round 6) against
    geometry.project_points(projection="fma") on the device -- the torch op sequence of point_sampling_trt
    (modules/encoder.py:197-259) that test_base_projection_on_device_is_bit_exact pins to the reference's CPU arrays:
    fp32 results bit for bit, fp16 results = the fp32 results rounded once."""
    import bevformer_tensorrt_amd as bevops
    ref_3d = G.reference_points_3d(bev[0], bev[1], PC_RANGE[5] - PC_RANGE[2], 4, device="cpu")
    pillars = G.pillar_points(ref_3d, PC_RANGE).cuda()
    for l2i in _rigs(*image):
        l2i = l2i.cuda()
        cam, mask = G.project_points(pillars, l2i, image, projection="fma")
        got_cam, got_mask = bevops.point_sampling(pillars, l2i, image, torch.float32)
        assert _same_bits(got_cam, cam.contiguous()) and _same_bits(got_mask, mask.contiguous())
        h_cam, h_mask = bevops.point_sampling(pillars, l2i, image, torch.float16)
        assert _same_bits(h_cam, cam.half().contiguous()) and _same_bits(h_mask, mask.half().contiguous())
        assert 0.02 < float((mask > 0).float().mean()) < 0.9          # the case is not degenerate


def test_point_sampling_kernel_reproduces_the_reference_arrays():
    """... and straight against the golden: SHA-256 of the arrays the reference's own point_sampling_trt produced on
    the CPU at the base size (tests/golden/make_wrapper_golden.py)."""
    import bevformer_tensorrt_amd as bevops
    g = golden("geometry_base")
    bh, bw, ih, iw, step = (int(x) for x in g["meta"])
    ref_3d = G.reference_points_3d(bh, bw, PC_RANGE[5] - PC_RANGE[2], 4, device="cpu")
    pillars = G.pillar_points(ref_3d, PC_RANGE).cuda()
    cam, mask = bevops.point_sampling(pillars, torch.from_numpy(g["lidar2img"]).cuda(), (ih, iw), torch.float32)
    cam, mask = cam.cpu().numpy(), mask.cpu().numpy()
    assert np.array_equal(cam[:, :, ::step], g["cam_sample"])
    assert np.array_equal(mask[:, ::step], g["mask_sample"])
    assert [_digest(cam), _digest(mask)] == list(g["sha256"][1:])


def test_model_projection_uses_the_kernel_and_builds_the_plan_of_its_cameras():
    """BEVFormer.project: the kernel's results (equal to the torch path's), and for the base pyramid the visibility plan
    of the cameras asked for -- a camera-sharded rank's plan lists ITS cameras (the size the planned sampler insists on)."""
    from bevformer_tensorrt_amd import bevformer as B
    from bevformer_tensorrt_amd.utils import lib as L
    model = B.BEVFormer("base", seed=0).cuda().half()
    H, W = B.CONFIGS["base"]["image"]
    l2i = G.synthetic_lidar2img((H, W)).cuda()
    ref_cam, bev_mask, plan = model.project(l2i, (H, W), torch.float16)
    cam, mask = G.project_points(model._static[2], l2i, (H, W), projection="fma")
    assert _same_bits(ref_cam, cam.half().contiguous()) and _same_bits(bev_mask, mask.half().contiguous())
    h = L.load_library()
    assert plan.numel() == h.bevops_sca_plan_size(6, 40000)
    _, _, plan2 = model.project(l2i, (H, W), torch.float16, cams=[1, 4])
    assert plan2.numel() == h.bevops_sca_plan_size(2, 40000)
    counts = plan2[:64].view(torch.int32)[:2].cpu()
    assert [int(c) for c in counts] == [int((bev_mask[c] != 0).sum()) for c in (1, 4)]
    tiny = B.BEVFormer("tiny", seed=0).cuda().half()
    assert len(tiny.project(G.synthetic_lidar2img((480, 800)).cuda(), (480, 800), torch.float16)) == 2   # no plan: one level
