"""Index / grid generation feeding the samplers.  These are
the torch-level host computations of the reference's `*TRTP` wrappers, restated with the
same torch ops in the same order so that they are bit-identical on the same backend:

  reference_points_3d / _2d : BEVFormerEncoderTRTP.get_reference_points_3d and the
                              ref_2d slice (det2trt/models/modules/encoder.py:170-195,291)
  point_sampling            : BEVFormerEncoderTRTP.point_sampling_trt (encoder.py:197-259)
                              -> reference_points_cam [cams,1,nq,D,2], bev_mask [cams,nq,1]
  bev_shift                 : ego-motion shift of the TSA grid
                              (det2trt/models/modules/transformer.py:262-294)
  hybrid_ref_2d             : encoder.py:297-307
  refine_reference_points   : decoder reference-point refinement (decoder.py:24-40,93-103)
  level_layout              : spatial_shapes / level_start_index (transformer.py:313-321,
                              functions/multi_scale_deformable_attn.py:103-106)
  synthetic_lidar2img       : a 6-camera ring rig for synthetic frames (new; the reference
                              takes lidar2img from the nuScenes dataset)
Every address the SCA/TSA samplers touch derives from these values.
"""
import math

import numpy as np
import torch


def reference_points_3d(H, W, Z=8, num_points_in_pillar=4, device="cuda", dtype=torch.float):
    zs = (torch.linspace(0.5, Z - 0.5, num_points_in_pillar, dtype=dtype, device=device)
          .view(-1, 1, 1).repeat(1, H, W) / Z)
    xs = (torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device)
          .view(1, 1, W).repeat(num_points_in_pillar, H, 1) / W)
    ys = (torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device)
          .view(1, H, 1).repeat(num_points_in_pillar, 1, W) / H)
    return torch.stack((xs, ys, zs), -1).view(1, num_points_in_pillar, -1, 3)


def reference_points_2d(ref_3d):
    return ref_3d[0, 0, :, :2].view(1, -1, 1, 2).clone()


def point_sampling(reference_points, pc_range, lidar2img, image_shape, num_cams=6, projection="matmul"):
    """reference_points [1,D,nq,3] in [0,1]^3; lidar2img [*,num_cams,4,4]; image_shape (h, w).
    Returns reference_points_cam [num_cams,1,nq,D,2] (normalised image coords) and
    bev_mask [num_cams,nq,1] = visible / max(#cameras seeing the pillar, 1e-4).
    projection="matmul" is the reference's statement (torch.matmul of 960 000 4x4 @ 4x1 products,
    encoder.py:223: one degenerate batched GEMM, 12.6 ms on MI355X at base); "fma" evaluates
    the same 4-term dot products with broadcast multiply + sum (fp32, k ascending) -- equal to
    the BLAS result to the last ulp or two and ~100x faster; the model uses "fma"."""
    pts = pillar_points(reference_points, pc_range)
    return project_points(pts, lidar2img, image_shape, num_cams, projection)


def pillar_points(reference_points, pc_range):
    """First half of point_sampling_trt (encoder.py:199-219): [0,1]^3 pillar anchors -> metric
    homogeneous points [D,1,1,nq,4,1].  Frame-independent (a model may cache it)."""
    D = reference_points.shape[1]
    scale = torch.tensor([pc_range[3] - pc_range[0], pc_range[4] - pc_range[1],
                          pc_range[5] - pc_range[2]], dtype=reference_points.dtype,
                         device=reference_points.device).view(1, 1, 1, 3)
    origin = torch.tensor(pc_range[:3], dtype=reference_points.dtype,
                          device=reference_points.device)
    pts = reference_points * scale + origin
    pts = torch.cat((pts, torch.ones_like(pts[..., :1])), -1)
    return pts.view(D, 1, 1, -1, 4, 1)


def project_points(pts, lidar2img, image_shape, num_cams=6, projection="matmul"):
    """Second half of point_sampling_trt (encoder.py:220-259): projection, normalisation, mask.
    Bit-exactness across devices: projection="fma" spells the 4-term dot product as separately
    rounded multiplies and adds in ascending k -- what the CPU BLAS does for these 4x4 @ 4x1
    products (tests/test_geometry_cpu.py holds them equal at the base size) -- and the image
    size divides as a TENSOR: with a Python scalar the GPU division kernel multiplies by the
    rounded reciprocal, which is not the reference's CPU result for 1600 or 928."""
    D = pts.shape[0]
    l2i = lidar2img.view(1, 1, num_cams, 1, 4, 4)
    if projection == "matmul":
        cam = torch.matmul(l2i, pts).squeeze(-1)
    else:
        p = pts.squeeze(-1).unsqueeze(-2)                    # [D,1,1,nq,1,4]
        cam = l2i[..., 0] * p[..., 0]
        for k in range(1, 4):
            cam = cam + l2i[..., k] * p[..., k]
    eps = 1e-5
    zeros = cam.new_zeros(D, 1, num_cams, int(cam.shape[3]), 1, dtype=torch.float32)
    ones = zeros + 1
    bev_mask = torch.where(cam[..., 2:3] > eps, ones, zeros)
    cam = cam[..., 0:2] / torch.max(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
    # (torch.full is a fill kernel: legal under stream capture, unlike a host-to-device copy)
    cam[..., 0] /= torch.full((), float(image_shape[1]), dtype=cam.dtype, device=cam.device)
    cam[..., 1] /= torch.full((), float(image_shape[0]), dtype=cam.dtype, device=cam.device)
    bev_mask *= torch.where(cam[..., 1:2] > 0.0, ones, zeros)
    bev_mask *= torch.where(cam[..., 1:2] < 1.0, ones, zeros)
    bev_mask *= torch.where(cam[..., 0:1] < 1.0, ones, zeros)
    bev_mask *= torch.where(cam[..., 0:1] > 0.0, ones, zeros)
    cam = cam.permute(2, 1, 3, 0, 4)
    bev_mask = (1 - (1 - bev_mask).prod(0)).view(num_cams, -1, 1)
    bev_mask = bev_mask / torch.clamp(bev_mask.sum(0, keepdims=True), min=1e-4)
    return cam, bev_mask


def refine_reference_points(tmp, reference_points):
    """Decoder reference-point refinement (det2trt/models/modules/decoder.py:93-103):
    tmp [1, nq, 10] = reg_branch output, reference_points [1, nq, 3] in (0, 1).  The decoder
    uses ITS OWN inverse_sigmoid (decoder.py:24-40: one clamp to [eps, 1-eps]); the head uses
    mmdet's (bevformer_head.py:6,254: clamp to [0, 1], then each factor to >= eps) -- they differ
    below eps, so both are restated."""
    # (x, y) and z refined in one pass over [.., 3] (element-wise: the same values as the reference's two slices)
    return (torch.cat([tmp[..., :2], tmp[..., 4:5]], dim=-1) + inverse_sigmoid_decoder(reference_points)).sigmoid()


def inverse_sigmoid_decoder(x, eps=1e-5):
    """det2trt/models/modules/decoder.py:24-40."""
    x = x.clamp(min=eps, max=1 - eps)
    return torch.log(x / (1 - x))


def inverse_sigmoid(x, eps=1e-5):
    """mmdet.models.utils.transformer.inverse_sigmoid (mmdet 2.25.1, the form
    third_party/bev_mmdet3d/models/modules/decoder.py:38-54 restates), used by the head."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def bev_shift(can_bus, bev_h, bev_w, grid_length=(0.512, 0.512), use_shift=True):
    """can_bus [18]: [0:2] = ego translation (m), [-2] = ego yaw (rad), [-1] = rotation (deg).
    Returns shift [1,2] = (shift_x, shift_y) in normalised BEV units."""
    delta_x, delta_y = can_bus[0:1], can_bus[1:2]
    ego_angle = can_bus[-2:-1] / np.pi * 180
    grid_length_y, grid_length_x = grid_length[0], grid_length[1]
    translation_length = torch.sqrt(delta_x ** 2 + delta_y ** 2)
    translation_angle = ((torch.atan(delta_y / (delta_x + 1e-8))
                          + ((1 - torch.sign(delta_x)) / 2) * torch.sign(delta_y) * np.pi)
                         / np.pi * 180)
    bev_angle = ego_angle - translation_angle
    shift_y = translation_length * torch.cos(bev_angle / 180 * np.pi) / grid_length_y / bev_h
    shift_x = translation_length * torch.sin(bev_angle / 180 * np.pi) / grid_length_x / bev_w
    shift_y = shift_y * int(use_shift)
    shift_x = shift_x * int(use_shift)
    return torch.cat([shift_x, shift_y]).unsqueeze(0)


def hybrid_ref_2d(ref_2d, shift, use_prev_bev):
    shift_ref_2d = ref_2d.clone() + shift.view(1, 1, 1, 2) * use_prev_bev
    return torch.cat([shift_ref_2d, ref_2d], dim=0)


def level_layout(level_hw, device="cpu"):
    """[(h, w), ...] -> spatial_shapes [L,2] int64 (h, w), level_start_index [L] int64."""
    spatial_shapes = torch.tensor([[int(h), int(w)] for h, w in level_hw], dtype=torch.long,
                                  device=device)
    start = torch.zeros_like(spatial_shapes[:, 0])
    start[1:] = torch.cumsum(spatial_shapes[:, 0] * spatial_shapes[:, 1], dim=0)[:-1]
    return spatial_shapes, start


def synthetic_lidar2img(image_hw=(928, 1600), yaws_deg=(0.0, 55.0, -55.0, 180.0, 110.0, -110.0),
                        focal=(1260.0, 1260.0, 1260.0, 810.0, 1260.0, 1260.0), cam_height=1.5,
                        radius=1.0, dtype=torch.float32):
    """6-camera ring rig (front, front-left, front-right, back, back-left, back-right) in the
    nuScenes arrangement; lidar frame x forward / y left / z up, origin at sensor height.
    Returns lidar2img [1,6,4,4] such that [u*z, v*z, z, 1] = lidar2img @ [x, y, z, 1]."""
    h, w = image_hw
    mats = []
    for yaw, f in zip(yaws_deg, focal):
        a = math.radians(yaw)
        fwd = np.array([math.cos(a), math.sin(a), 0.0])           # camera looks along fwd
        left = np.array([-math.sin(a), math.cos(a), 0.0])
        up = np.array([0.0, 0.0, 1.0])
        R = np.stack([-left, -up, fwd])                            # cam axes: x right, y down, z fwd
        t = fwd * radius + np.array([0.0, 0.0, cam_height - 1.84])  # camera centre in lidar frame
        ext = np.eye(4)
        ext[:3, :3] = R
        ext[:3, 3] = -R @ t
        s = w / 1600.0
        K = np.eye(4)
        K[0, 0] = K[1, 1] = f * s
        K[0, 2], K[1, 2] = w / 2.0 + 3.0, h / 2.0 - 12.0
        mats.append(K @ ext)
    return torch.tensor(np.stack(mats)[None], dtype=dtype)
