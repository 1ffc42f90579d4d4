"""Pure-torch stand-ins for the sampling operators, used ONLY by tests to check the re-hosted
model's dataflow: the same network evaluated with the reference's PyTorch formulations
(oracle/torch_ref.py MSDA, grid_sampler-based rotate as functions/rotate.py:12-80)."""
import math

import torch
import torch.nn.functional as F

from . import torch_ref


class RefOps:
    @staticmethod
    def multi_scale_deformable_attn(value, shapes, ref, off, w):
        shapes = shapes.to(value.device).long()
        out = torch_ref.msda(value.float(), shapes, ref.float(), off.float().contiguous(), w.float().contiguous())
        return out.to(value.dtype)

    @staticmethod
    def rotate(img, angle, center, interpolation="nearest"):
        C, H, W = img.shape
        a = -float(angle) * math.pi / 180
        cx, cy = float(center[0]) - W * 0.5, float(center[1]) - H * 0.5
        cs, sn = math.cos(a), math.sin(a)
        theta = torch.tensor([[cs, sn, -cx * cs - cy * sn + cx], [-sn, cs, cx * sn - cy * cs + cy]],
                             dtype=torch.float32, device=img.device)
        xs = torch.linspace(-W * 0.5 + 0.5, W * 0.5 - 0.5, W, device=img.device)
        ys = torch.linspace(-H * 0.5 + 0.5, H * 0.5 - 0.5, H, device=img.device)
        base = torch.stack([xs.expand(H, W), ys.unsqueeze(1).expand(H, W), torch.ones(H, W, device=img.device)], -1)
        rt = 2 * theta.t()
        rt[:, 0] /= W
        rt[:, 1] /= H
        grid = (base.view(-1, 3) @ rt).view(1, H, W, 2)
        out = F.grid_sample(img.float()[None], grid, mode=interpolation, padding_mode="zeros", align_corners=False)
        return out[0].to(img.dtype)

    @staticmethod
    def modulated_deformable_conv2d(x, offset, mask, weight, bias, stride, padding, dilation, groups, dg):
        import oracle
        out = oracle.mdconv(x.float().cpu().numpy(), offset.float().cpu().numpy(), mask.float().cpu().numpy(),
                            weight.float().cpu().numpy(), None if bias is None else bias.float().cpu().numpy(),
                            (stride,) * 2, (padding,) * 2, (dilation,) * 2, groups, dg)
        return torch.from_numpy(out).to(x.device, x.dtype)
