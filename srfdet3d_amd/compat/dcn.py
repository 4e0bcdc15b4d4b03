"""Modulated deformable convolution (DCNv2) for the one config that asks for it: the ResNet-101 image backbone of
configs/others/srfdet_dvoxel_waymo_LC.py (`dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False)`,
`stage_with_dcn=(False, False, True, True)`).

In the reference this is mmcv's `ModulatedDeformConv2dPack` (a CUDA operator outside the reference tree); mmcv is not
available here, so nothing pins this module against it ("parity unpinned"); it follows the published definition
  y(p) = sum_k w_k . x(p + p_k + dp_k) . m_k       (Zhu et al., "Deformable ConvNets v2", eq. 1)
with bilinear sampling and zero padding, and mmcv's tensor conventions: `conv_offset` yields 3.K.G channels, split in three
chunks (o1, o2, mask); offset = cat(o1, o2) is read as interleaved (dy, dx) pairs per kernel tap; mask = sigmoid(mask).
Written on torch ops (`grid_sample` per tap), i.e. differentiable and device-agnostic; this branch is not on the measured
path.  Parameter names follow mmcv (`weight`, `bias`, `conv_offset.weight`, `conv_offset.bias`).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1, groups=1, deform_groups=1):
    """x (N,C,H,W); offset (N, 2.K.G, Ho, Wo); mask (N, K.G, Ho, Wo); weight (Cout, C/groups, kh, kw) -> (N, Cout, Ho, Wo)."""
    N, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    K, G = kh * kw, deform_groups
    Ho, Wo = offset.shape[-2:]
    assert offset.shape[1] == 2 * K * G and mask.shape[1] == K * G and C % G == 0
    dev, dt = x.device, x.dtype
    base_y = (torch.arange(Ho, device=dev, dtype=dt) * stride - padding).view(1, Ho, 1)
    base_x = (torch.arange(Wo, device=dev, dtype=dt) * stride - padding).view(1, 1, Wo)
    sx = 2.0 / max(W - 1, 1)
    sy = 2.0 / max(H - 1, 1)
    Cg = C // G
    out = None
    for t in range(K):
        i, j = divmod(t, kw)
        cols = []
        for g in range(G):
            py = base_y + i * dilation + offset[:, g * 2 * K + 2 * t]
            px = base_x + j * dilation + offset[:, g * 2 * K + 2 * t + 1]
            grid = torch.stack([px * sx - 1.0, py * sy - 1.0], dim=-1)
            s = F.grid_sample(x[:, g * Cg:(g + 1) * Cg], grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            cols.append(s * mask[:, g * K + t].unsqueeze(1))
        col = cols[0] if G == 1 else torch.cat(cols, dim=1)
        y = F.conv2d(col, weight[:, :, i:i + 1, j:j + 1], None, 1, 0, 1, groups)
        out = y if out is None else out + y
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


class ModulatedDeformConv2dPack(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=True):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, (k, k)
        self.stride, self.padding, self.dilation, self.groups, self.deform_groups = stride, padding, dilation, groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 3 * k * k, k, stride, padding, dilation, bias=True)
        stdv = 1.0 / math.sqrt(in_channels * k * k)
        nn.init.uniform_(self.weight, -stdv, stdv)
        nn.init.zeros_(self.conv_offset.weight)   # zero offsets and mask logits at the start: a plain conv scaled by 0.5
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        o1, o2, m = torch.chunk(self.conv_offset(x), 3, dim=1)
        return modulated_deform_conv2d(x, torch.cat((o1, o2), dim=1), torch.sigmoid(m), self.weight, self.bias, self.stride,
                                       self.padding, self.dilation, self.groups, self.deform_groups)
