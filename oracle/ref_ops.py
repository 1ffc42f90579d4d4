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


def mdconv_torch(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """Modulated deformable convolution (DCNv2) in plain torch fp32, groups = deform_groups = 1: the arithmetic of
    TensorRT/plugin/modulated_deformable_conv2d/modulatedDeformableConv2dKernel.cu:259-318 (tap position = ho * stride
    - pad + i * dilation + offset_h, likewise w; offsets laid out [2 * K * K, Ho, Wo] with h before w per tap; value =
    4-corner bilinear with per-corner bounds, zero outside, times the mask) and :695-760 (columns x weights + bias),
    with F.grid_sample(align_corners=True, zeros padding) as the bilinear tap.  Exists so that the model-level parity
    tests of the big configs can evaluate the reference formulation ON THE DEVICE in seconds (`RefOps` sends every DCNv2
    call through the host C oracle: 1.4 s per camera image at 58 x 100); tests/test_mdconv_cpu.py pins it against
    oracle.mdconv."""
    B, Cin, H, W = x.shape
    Cout, _, K, _ = weight.shape
    Ho = (H + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    x, offset, mask, weight = x.float(), offset.float(), mask.float(), weight.float()
    hs = (torch.arange(Ho, device=x.device, dtype=torch.float32) * stride - padding).view(1, Ho, 1)
    ws = (torch.arange(Wo, device=x.device, dtype=torch.float32) * stride - padding).view(1, 1, Wo)
    out = x.new_zeros(B, Cout, Ho, Wo)
    for i in range(K):
        for j in range(K):
            t = i * K + j
            h_im = hs + i * dilation + offset[:, 2 * t]
            w_im = ws + j * dilation + offset[:, 2 * t + 1]
            grid = torch.stack((2 * w_im / max(W - 1, 1) - 1, 2 * h_im / max(H - 1, 1) - 1), dim=-1)
            tap = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True) * mask[:, t:t + 1]
            out += torch.einsum("oc,bchw->bohw", weight[:, :, i, j], tap)
    return out if bias is None else out + bias.float().view(1, -1, 1, 1)


class TorchRefOps(RefOps):
    """RefOps with DCNv2 evaluated in torch on the tensors' own device (mdconv_torch) instead of the host C oracle."""

    @staticmethod
    def modulated_deformable_conv2d(x, offset, mask, weight, bias, stride, padding, dilation, groups, dg):
        assert groups == 1 and dg == 1
        return mdconv_torch(x, offset, mask, weight, bias, stride, padding, dilation).to(x.dtype)
