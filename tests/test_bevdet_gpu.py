"""BEVDet view-transform slice on the GPU (det2trt/models/detector/bevdet.py:50-76): depth_net ->
softmax -> bev_pool_v2 (HIP) -> [1, C, 128, 128], against the same slice with the oracle pooling
(torch.index_add_ in fp64) on the ranks the reference's own geometry produces."""
import numpy as np
import pytest
import torch

from conftest import golden
from util_bevpool import index_add_reference

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 2e-2)])
def test_view_transform_slice(dtype, tol):
    from bevformer_tensorrt_amd.bevdet import BEVDET_R50, LSSViewTransformer
    g = golden("bevdet_geometry")
    vt = LSSViewTransformer(**BEVDET_R50)
    t = lambda k: torch.from_numpy(g[k])
    ranks = vt.get_bev_pool_input(t("sensor2ego"), None, t("cam2imgs"), t("post_rots"), t("post_trans"), t("bda"))
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(6, 256, 16, 44, generator=gen)
    vt = vt.cuda().to(dtype)
    out = vt.view_transform(x.cuda().to(dtype), *[r.cuda() for r in ranks])
    assert out.shape == (1, 64, 128, 128)
    with torch.no_grad():
        y = vt.depth_net(x.cuda().to(dtype)).float().cpu()
    depth = y[:, :59].softmax(dim=1).to(dtype).float().numpy()
    feat = y[:, 59:123].permute(0, 2, 3, 1).contiguous().to(dtype).float().numpy()
    rb, rd, rf = (r.numpy() for r in ranks[:3])
    want = index_add_reference(depth, feat, rd, rf, rb, 128, 128).transpose(0, 3, 1, 2)
    err = np.abs(out.float().cpu().numpy() - want)
    assert err.max() <= tol * max(1.0, np.abs(want).max()), err.max()


class _OraclePoolOps:
    """Operator namespace of the REFERENCE data path of the whole detector: library convolutions (no fused entries
    here, so bevdet._conv takes F.conv2d) and the oracle statement of bev_pool_v2 (torch.index_add_, fp64)."""

    @staticmethod
    def bev_pool_v2_2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, bev_h, bev_w):
        want = index_add_reference(depth.float().cpu().numpy(), feat.float().cpu().numpy(), ranks_depth.cpu().numpy(),
                                   ranks_feat.cpu().numpy(), ranks_bev.cpu().numpy(), bev_h, bev_w)
        return torch.from_numpy(want).to(depth.device, depth.dtype)


def test_whole_detector_frame_matches_the_reference_path():
    """BEVDetTRT.forward_trt (det2trt/models/detector/bevdet.py:29-82) end to end: R50 + CustomFPN + depth_net +
    bev_pool_v2 + CustomResNet bev encoder + FPN_LSS + CenterHead.  The fp16 channels-last frame on this package's
    kernels (HIP bev_pool_v2, MFMA convolutions) against the SAME weights in fp32 through the library convolutions and
    the oracle pooling, on the calibration of the reference's own bev_pool test."""
    from bevformer_tensorrt_amd.bevdet import BEVDet
    g = golden("bevdet_geometry")
    t = lambda k: torch.from_numpy(g[k])
    fast = BEVDet(seed=0).cuda().half()
    ref = BEVDet(ops=_OraclePoolOps, seed=0).cuda().float()
    ref.load_state_dict({k: v.float() for k, v in fast.state_dict().items()})
    ranks = fast.view.get_bev_pool_input(t("sensor2ego"), None, t("cam2imgs"), t("post_rots"), t("post_trans"), t("bda"))
    ranks = [r.cuda() for r in ranks]
    img = torch.randn(1, 6, 3, 256, 704, generator=torch.Generator().manual_seed(1)).cuda()
    got = fast(img.half(), *ranks)
    want = ref(img, *ranks)
    names = ("reg", "height", "dim", "rot", "vel", "heatmap")
    for n, a, b in zip(names, got, want):
        assert a.shape == b.shape == (1, b.shape[1], 128, 128) and a.dtype == torch.float16
        err = (a.float() - b).abs().max().item()
        assert err <= 3e-2 * max(1.0, b.abs().max().item()), (n, err, b.abs().max().item())
    # and the frame replays from a HIP graph
    static = img.half().clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fast(static, *ranks)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = fast(static, *ranks)
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, got):
        assert (a.float() - b.float()).abs().max().item() <= 1e-2 * max(1.0, b.float().abs().max().item())


def test_int8_bevdet_engine_tracks_fp16():
    """The PTQ build of the whole BEVDet-R50 (quantization.build_int8_bevdet: ResNet-50 + FPN laterals as the int8
    activation chain, bev_pool_v2 on its INT8 plugin flavour) against the fp16 detector on an unseen frame."""
    from bevformer_tensorrt_amd import bevdet as D
    from bevformer_tensorrt_amd.quantization import build_int8_bevdet
    dev = torch.device("cuda")
    ref = D.BEVDet(seed=0).to(dev, torch.float16)
    ranks = [r.to(dev) for r in ref.view.get_bev_pool_input(*D.synthetic_rig(ref.view))]
    g = torch.Generator().manual_seed(2)
    frames = [(torch.randn(1, 6, 3, 256, 704, generator=g).to(dev, torch.float16), *ranks) for _ in range(3)]
    model, qops, note = build_int8_bevdet(D, dev, frames[:2])
    assert note["activation_chain"] and note["int8_plugin_sites"] == 1 and model.int8_chain.ready
    got, want = model(*frames[2]), ref(*frames[2])
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.isfinite(a.float()).all()
        rel = ((a.float() - b.float()).abs().mean() / b.float().abs().mean().clamp(min=1e-3)).item()
        assert rel <= 0.25, rel          # 8-bit noise of a random-weight network with unnormalised residual streams
