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
