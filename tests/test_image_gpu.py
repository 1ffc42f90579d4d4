"""bevops_image_normalize_pad (the camera-image front end of the frame loop) against
the numpy restatement of the reference pipeline (oracle/image_ref.py): fp32 output bit-exact,
fp16 output == the RNE cast of it; nuScenes size 900x1600 -> 928x1600 and an odd size; both layouts;
FrameRunner.step_raw == step on the pre-processed frame."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(900, 1600), (45, 70)])
@pytest.mark.parametrize("src", [torch.uint8, torch.float32])
@pytest.mark.parametrize("to_rgb", [False, True])
def test_matches_reference_pipeline(hw, src, to_rgb):
    import bevformer_tensorrt_amd as bev
    from oracle.image_ref import image_normalize_pad as ref
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (6, hw[0], hw[1], 3), generator=g, dtype=torch.uint8)
    if src == torch.float32:
        img = img.float() + torch.rand(img.shape, generator=g) * 0.5
    std = (58.395, 57.12, 57.375) if to_rgb else (1.0, 1.0, 1.0)
    want = ref(img.numpy(), std=std, to_rgb=to_rgb)
    got = bev.image_normalize_pad(img.cuda(), std=std, to_rgb=to_rgb, dtype=torch.float32)
    assert got.shape == want.shape == (6, 3) + bev.padded_size(*hw)
    assert np.array_equal(got.cpu().numpy(), want)
    for cl in (False, True):
        h = bev.image_normalize_pad(img.cuda(), std=std, to_rgb=to_rgb, dtype=torch.float16, channels_last=cl)
        assert h.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        assert torch.equal(h.cpu(), torch.from_numpy(want).half())


def test_frame_runner_step_raw():
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    import bevformer_tensorrt_amd as bev
    dev = torch.device("cuda")
    model = B.BEVFormer("tiny", seed=0).to(dev, torch.float16)
    H, W = B.CONFIGS["tiny"]["image"]
    g = torch.Generator().manual_seed(0)
    raw = torch.randint(0, 256, (6, H - 30, W, 3), generator=g, dtype=torch.uint8)   # 450x800 -> 480x800
    l2i = G.synthetic_lidar2img((H, W))
    a = B.FrameRunner(model, dev, torch.float16)
    b = B.FrameRunner(model, dev, torch.float16)
    cls_a, crd_a = a.step_raw(raw.to(dev), torch.zeros(18), l2i, "s")
    pre = bev.image_normalize_pad(raw.to(dev), dtype=torch.float16)[None]
    cls_b, crd_b = b.step(pre, torch.zeros(18), l2i, "s")
    # same frame through both entries; the dense library kernels (hipBLASLt / MIOpen) may pick other
    # algorithms on a second runner, so the bar is fp16 noise, not bit equality
    # (measured: 1.4e-2 on class logits, 0.25 on box coordinates of +-51 m between two runs of the SAME frame)
    assert (cls_a.float() - cls_b.float()).abs().max().item() <= 5e-2
    assert (crd_a.float() - crd_b.float()).abs().max().item() <= 0.5
    assert torch.equal(a._in["image"], pre)       # the pre-processing itself is bit-identical
