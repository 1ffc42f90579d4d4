"""BEVDet's view transformer on the MI355X operators (BASELINE config 5).

`LSSViewTransformer` restates the pieces of the reference's Lift-Splat-Shoot neck that surround
the bev_pool_v2 plugin:

  * frustum / calibration geometry -- create_grid_infos, create_frustum, get_lidar_coor,
    voxel_pooling_prepare_v2 (third_party/bev_mmdet3d/models/necks/view_transformer.py:66-168,
    239-312; det2trt/models/necks/view_transformer.py:8 `LSSViewTransformerTRT` adds nothing to
    them).  Index generation must be BIT-EXACT: same torch ops in the same order; the golden
    (tests/golden/bevdet_geometry.npz) is produced by executing the reference's own methods on the
    calibration the reference's test hard-codes.  The ranks are computed once per rig on the HOST
    (the reference feeds them to the engine as inputs, tools/bevdet/evaluate_trt.py:107-127);
  * `view_transform` -- the slice of BEVDetTRT.forward_trt between the image neck and the BEV
    encoder (det2trt/models/detector/bevdet.py:50-76): depth_net (1x1 conv) -> softmax over the D
    depth bins -> bev_pool_v2 (HIP) -> [B, C, bev_h, bev_w].

BEVDet-R50 config (configs/bevdet/bevdet-r50-cbgs.py:44-104): 6 cameras of 256x704, downsample 16
(16x44 features), 59 depth bins, 64 channels, 128x128 BEV cells of 0.8 m.
"""
import torch
import torch.nn as nn

from . import functions as _hip_ops

BEVDET_R50 = dict(
    grid_config=dict(x=[-51.2, 51.2, 0.8], y=[-51.2, 51.2, 0.8], z=[-5, 3, 8], depth=[1.0, 60.0, 1.0]),
    input_size=(256, 704), downsample=16, in_channels=256, out_channels=64)


class LSSViewTransformer(nn.Module):
    def __init__(self, grid_config, input_size, downsample, in_channels, out_channels, ops=None, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.ops = ops if ops is not None else _hip_ops
        self.create_grid_infos(**grid_config)
        self.frustum = self.create_frustum(grid_config["depth"], input_size, downsample)
        self.out_channels, self.in_channels = out_channels, in_channels
        self.depth_net = nn.Conv2d(in_channels, self.D + out_channels, kernel_size=1, padding=0)   # :59-61

    # ---- view_transformer.py:66-83
    def create_grid_infos(self, x, y, z, **kwargs):
        self.grid_lower_bound = torch.Tensor([cfg[0] for cfg in [x, y, z]])
        self.grid_interval = torch.Tensor([cfg[2] for cfg in [x, y, z]])
        self.grid_size = torch.Tensor([(cfg[1] - cfg[0]) / cfg[2] for cfg in [x, y, z]])

    # ---- view_transformer.py:85-124 (sid = False)
    def create_frustum(self, depth_cfg, input_size, downsample):
        H_in, W_in = input_size
        H_feat, W_feat = H_in // downsample, W_in // downsample
        d = torch.arange(*depth_cfg, dtype=torch.float).view(-1, 1, 1).expand(-1, H_feat, W_feat)
        self.D = d.shape[0]
        x = torch.linspace(0, W_in - 1, W_feat, dtype=torch.float).view(1, 1, W_feat).expand(self.D, H_feat, W_feat)
        y = torch.linspace(0, H_in - 1, H_feat, dtype=torch.float).view(1, H_feat, 1).expand(self.D, H_feat, W_feat)
        return torch.stack((x, y, d), -1)

    # ---- view_transformer.py:126-168
    def get_lidar_coor(self, sensor2ego, ego2global, cam2imgs, post_rots, post_trans, bda):
        B, N, _, _ = sensor2ego.shape
        points = self.frustum.to(sensor2ego) - post_trans.view(B, N, 1, 1, 1, 3)
        points = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(points.unsqueeze(-1))
        points = torch.cat((points[..., :2, :] * points[..., 2:3, :], points[..., 2:3, :]), 5)
        combine = sensor2ego[:, :, :3, :3].matmul(torch.inverse(cam2imgs))
        points = combine.view(B, N, 1, 1, 1, 3, 3).matmul(points).squeeze(-1)
        points += sensor2ego[:, :, :3, 3].view(B, N, 1, 1, 1, 3)
        points = bda.view(B, 1, 1, 1, 1, 3, 3).matmul(points.unsqueeze(-1)).squeeze(-1)
        return points

    # ---- view_transformer.py:239-312
    def voxel_pooling_prepare_v2(self, coor):
        B, N, D, H, W, _ = coor.shape
        num_points = B * N * D * H * W
        ranks_depth = torch.arange(0, num_points, dtype=torch.int, device=coor.device)
        ranks_feat = torch.arange(0, num_points // D, dtype=torch.int, device=coor.device)
        ranks_feat = ranks_feat.reshape(B, N, 1, H, W).expand(B, N, D, H, W).flatten()
        coor = (coor - self.grid_lower_bound.to(coor)) / self.grid_interval.to(coor)
        coor = coor.long().view(num_points, 3)
        batch_idx = torch.arange(0, B).reshape(B, 1).expand(B, num_points // B).reshape(num_points, 1).to(coor)
        coor = torch.cat((coor, batch_idx), 1)
        kept = ((coor[:, 0] >= 0) & (coor[:, 0] < self.grid_size[0]) & (coor[:, 1] >= 0)
                & (coor[:, 1] < self.grid_size[1]) & (coor[:, 2] >= 0) & (coor[:, 2] < self.grid_size[2]))
        if len(kept) == 0:
            return None, None, None, None, None
        coor, ranks_depth, ranks_feat = coor[kept], ranks_depth[kept], ranks_feat[kept]
        ranks_bev = coor[:, 3] * (self.grid_size[2] * self.grid_size[1] * self.grid_size[0])
        ranks_bev += coor[:, 2] * (self.grid_size[1] * self.grid_size[0])
        ranks_bev += coor[:, 1] * self.grid_size[0] + coor[:, 0]
        order = ranks_bev.argsort()
        ranks_bev, ranks_depth, ranks_feat = ranks_bev[order], ranks_depth[order], ranks_feat[order]
        kept = torch.ones(ranks_bev.shape[0], device=ranks_bev.device, dtype=torch.bool)
        kept[1:] = ranks_bev[1:] != ranks_bev[:-1]
        interval_starts = torch.where(kept)[0].int()
        if len(interval_starts) == 0:
            return None, None, None, None, None
        interval_lengths = torch.zeros_like(interval_starts)
        interval_lengths[:-1] = interval_starts[1:] - interval_starts[:-1]
        interval_lengths[-1] = ranks_bev.shape[0] - interval_starts[-1]
        return (ranks_bev.int().contiguous(), ranks_depth.int().contiguous(), ranks_feat.int().contiguous(),
                interval_starts.int().contiguous(), interval_lengths.int().contiguous())

    def get_bev_pool_input(self, sensor2keyegos, ego2globals, intrins, post_rots, post_trans, bda):
        """BEVDetTRT.get_bev_pool_input (det2trt/models/detector/bevdet.py:14-27)."""
        coor = self.get_lidar_coor(sensor2keyegos, ego2globals, intrins, post_rots, post_trans, bda)
        return self.voxel_pooling_prepare_v2(coor)

    @torch.no_grad()
    def view_transform(self, x, ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths):
        """x [N_cams, in_channels, H_feat, W_feat] (image-neck output) -> BEV features
        [1, out_channels, bev_h, bev_w]: BEVDetTRT.forward_trt, det2trt/models/detector/bevdet.py:50-76."""
        x = self.depth_net(x)
        depth = x[:, : self.D].softmax(dim=1)
        tran_feat = x[:, self.D: self.D + self.out_channels].permute(0, 2, 3, 1)
        depth, tran_feat = depth.contiguous(), tran_feat.contiguous()
        bev_h, bev_w = int(self.grid_size[1]), int(self.grid_size[0])
        out = self.ops.bev_pool_v2_2(depth, tran_feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                     interval_lengths, bev_h, bev_w)
        return out.permute(0, 3, 1, 2).contiguous()
