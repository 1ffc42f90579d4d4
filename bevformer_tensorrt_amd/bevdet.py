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


# --------------------------------------------------------------------------- the whole detector (BASELINE config 5)
# BEVDetTRT.forward_trt (det2trt/models/detector/bevdet.py:29-82) re-hosted without mmcv / mmdet, frozen BatchNorm
# folded into the convolution in front of it:
#   image [1, 6, 3, 256, 704] -> ResNet-50 (style "pytorch", out_indices (2, 3); configs/bevdet/bevdet-r50-cbgs.py:73-84)
#   -> CustomFPN (1024 / 2048 -> 256, top-down add, ONE 3x3 output convolution on the stride-16 level;
#      third_party/bev_mmdet3d/models/necks/fpn.py) -> depth_net -> depth softmax -> bev_pool_v2 (the plugin)
#   -> CustomResNet bev encoder (basic blocks, 64 -> 128 / 256 / 512, strides 2; models/backbone/bev_resnet.py)
#   -> FPN_LSS (bilinear x4 of the stride-8 level, concatenation with the stride-2 level, two 3x3 convolutions,
#      bilinear x2, 3x3 + 1x1; models/necks/lss_fpn.py) -> CenterHead.forward_trt (shared 3x3 convolution, one task,
#      six two-convolution heads with final_kernel 3; det2trt/models/dense_heads/centerpoint_head.py:42-53).
# Two data paths through the same weights, as in bevformer.py: `forward` (NCHW, library convolutions -- the reference
# op sequence, any device / dtype) and the channels-last fp16 path on this package's convolution / GEMM kernels.
from . import bevformer as _B   # noqa: E402  (ResNet and the channels-last convolution helpers)
import torch.nn.functional as F   # noqa: E402

HEADS_R50 = (("reg", 2), ("height", 1), ("dim", 3), ("rot", 2), ("vel", 2), ("heatmap", 10))


def _conv(ops, x, conv, relu=False, residual=None):
    """act(conv(x) + bias + residual) for a BN-folded nn.Conv2d: channels-last fp16 tensors take this package's
    kernels (1x1: GEMM over the pixel rows; 3x3: implicit GEMM), everything else the library convolution."""
    fast = x.is_cuda and x.dtype == torch.float16 and x.is_contiguous(memory_format=torch.channels_last) \
        and hasattr(ops, "conv3x3_auto") and conv.out_channels % 8 == 0    # (the heads' 1 / 2 / 3-channel outputs: library)
    if fast and conv.kernel_size == (1, 1):
        return _B._conv1x1_nhwc(ops, x, conv, relu, residual)
    if fast and conv.kernel_size == (3, 3) and conv.in_channels % 32 == 0:
        return _B._conv_nhwc(ops, x, conv, relu, residual)
    y = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class BasicBlock(nn.Module):
    """mmdet BasicBlock with the 3x3 downsample convolution CustomResNet gives the first block of a stage."""

    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv1, self.conv2 = nn.Conv2d(cin, cout, 3, stride, 1), nn.Conv2d(cout, cout, 3, 1, 1)
        self.downsample = nn.Conv2d(cin, cout, 3, stride, 1) if downsample else None

    def forward(self, x, ops):
        idt = x if self.downsample is None else _conv(ops, x, self.downsample)
        return _conv(ops, _conv(ops, x, self.conv1, True), self.conv2, True, idt)


class CustomFPN(nn.Module):
    """CustomFPN(in_channels=[1024, 2048], out_channels=256, num_outs=1, out_ids=[0]) of the BEVDet-R50 config
    (third_party/bev_mmdet3d/models/necks/fpn.py:158-183): two lateral 1x1 convolutions, nearest top-down add, one
    3x3 output convolution on the finer level."""

    def __init__(self, cins=(1024, 2048), cout=256):
        super().__init__()
        self.lateral = nn.ModuleList(nn.Conv2d(c, cout, 1) for c in cins)
        self.fpn_conv = nn.Conv2d(cout, cout, 3, 1, 1)

    def topdown_nhwc(self, lats, ops):
        """everything behind the lateral convolutions (the INT8 chain evaluates those itself); any layout"""
        l4, l5 = lats
        up_add = getattr(ops, "upsample_add_nhwc_", None)
        if up_add is not None and l4.is_cuda and l4.dtype == torch.float16 \
                and l4.is_contiguous(memory_format=torch.channels_last) and l5.is_contiguous(memory_format=torch.channels_last):
            up_add(l4, l5)
        else:
            l4 = l4 + F.interpolate(l5, size=l4.shape[2:], mode="nearest")
        return _conv(ops, l4, self.fpn_conv)

    def forward(self, feats, ops):
        return self.topdown_nhwc([_conv(ops, f, l) for l, f in zip(self.lateral, feats)], ops)


class BEVDet(nn.Module):
    """forward(image [1, 6, 3, H, W], ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths)
    -> (reg, height, dim, rot, vel, heatmap), each [1, c, 128, 128] -- BEVDetTRT.forward_trt."""

    def __init__(self, cfg=None, ops=None, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        cfg = cfg or BEVDET_R50
        self.cfg = cfg
        self.ops = ops = ops if ops is not None else _hip_ops
        self.backbone = _B.ResNet(50, (False,) * 4, (2, 3), ops, "pytorch")
        self.neck = CustomFPN()
        self.view = LSSViewTransformer(**{k: cfg[k] for k in ("grid_config", "input_size", "downsample", "in_channels",
                                                              "out_channels")}, ops=ops, seed=seed)
        c = cfg["out_channels"]
        chans, cin, stages = [2 * c, 4 * c, 8 * c], c, []
        for ch in chans:
            stages.append(nn.ModuleList([BasicBlock(cin, ch, 2, True), BasicBlock(ch, ch, 1, False)]))
            cin = ch
        self.bev_stages = nn.ModuleList(stages)
        self.neck_conv = nn.ModuleList([nn.Conv2d(chans[2] + chans[0], 512, 3, 1, 1), nn.Conv2d(512, 512, 3, 1, 1)])
        self.up2_conv = nn.ModuleList([nn.Conv2d(512, 256, 3, 1, 1), nn.Conv2d(256, 256, 1)])
        self.shared_conv = nn.Conv2d(256, 64, 3, 1, 1)
        self.heads = nn.ModuleDict({k: nn.ModuleList([nn.Conv2d(64, 64, 3, 1, 1), nn.Conv2d(64, n, 3, 1, 1)])
                                    for k, n in HEADS_R50})
        self.eval()

    def image_features(self, image):
        """img_backbone + img_neck: [6, 3, H, W] -> [6, 256, H / 16, W / 16]."""
        ops = self.ops
        nhwc = image.is_cuda and image.dtype == torch.float16 and hasattr(ops, "conv3x3_auto")
        if nhwc:
            if not getattr(self, "_nhwc_ready", False):
                for m in self.modules():
                    if isinstance(m, nn.Conv2d) and m.kernel_size != (1, 1):
                        m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
                self._nhwc_ready = True
            chain = getattr(self, "int8_chain", None)     # quantization.Int8ChainBackbone, after its freeze()
            if chain is not None and chain.ready:
                return chain(image)
            feats = self.backbone.forward_nhwc(image, ops)
        else:
            feats = self.backbone(image)
        return self.neck(feats, ops)

    def bev_encoder(self, x):
        ops = self.ops
        if x.is_cuda and x.dtype == torch.float16:
            x = x.contiguous(memory_format=torch.channels_last)
        feats = []
        for stage in self.bev_stages:
            for blk in stage:
                x = blk(x, ops)
            feats.append(x)
        x1 = F.interpolate(feats[2], scale_factor=4, mode="bilinear", align_corners=True)
        x = torch.cat([feats[0], x1], dim=1)
        if feats[0].is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        x = _conv(ops, _conv(ops, x, self.neck_conv[0], True), self.neck_conv[1], True)
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        return _conv(ops, _conv(ops, x, self.up2_conv[0], True), self.up2_conv[1])

    @torch.no_grad()
    def forward(self, image, ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths):
        x = self.image_features(image.flatten(0, 1))
        bev = self.view.view_transform(x, ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths)
        feat = self.bev_encoder(bev)
        ops = self.ops
        s = _conv(ops, feat, self.shared_conv, True)
        return tuple(_conv(ops, _conv(ops, s, h[0], True), h[1]) for h in (self.heads[k] for k, _ in HEADS_R50))


def synthetic_rig(view, n_cams=6, seed=0):
    """A plausible six-camera calibration for timing and tests (no nuScenes data here): cameras on a ring, 60 degrees
    apart, looking outwards, the resize / crop augmentation of the test pipeline as post_rots / post_trans.
    Returns the six tensors of LSSViewTransformer.get_bev_pool_input."""
    import math
    H, W = view_input_size(view)
    s2e = torch.zeros(1, n_cams, 4, 4)
    for i in range(n_cams):
        yaw = math.radians(60.0 * i)
        # camera axes (x right, y down, z forward) in the ego frame (x forward, y left, z up)
        fwd = torch.tensor([math.cos(yaw), math.sin(yaw), 0.0])
        right = torch.tensor([math.sin(yaw), -math.cos(yaw), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        s2e[0, i, :3, 0], s2e[0, i, :3, 1], s2e[0, i, :3, 2] = right, down, fwd
        s2e[0, i, :3, 3] = torch.tensor([1.5 * math.cos(yaw), 1.5 * math.sin(yaw), 1.6])
        s2e[0, i, 3, 3] = 1.0
    e2g = torch.eye(4).view(1, 1, 4, 4).repeat(1, n_cams, 1, 1)
    K = torch.tensor([[1266.0, 0.0, 800.0], [0.0, 1266.0, 450.0], [0.0, 0.0, 1.0]]).view(1, 1, 3, 3).repeat(1, n_cams, 1, 1)
    scale = W / 1600.0
    post_rots = (torch.eye(3) * scale).view(1, 1, 3, 3).repeat(1, n_cams, 1, 1)
    post_rots[..., 2, 2] = 1.0
    post_trans = torch.zeros(1, n_cams, 3)
    post_trans[..., 1] = -(900.0 * scale - H)
    bda = torch.eye(3).view(1, 3, 3)
    return s2e, e2g, K, post_rots, post_trans, bda


def view_input_size(view):
    """(H, W) of the camera images the frustum of `view` was built for."""
    x, y = view.frustum[0, 0, :, 0], view.frustum[0, :, 0, 1]
    return int(round(float(y[-1]))) + 1, int(round(float(x[-1]))) + 1
