"""`naiveSyncBN1dCustom` (mmdet3d_plugin/ops/norm.py:27-84): BatchNorm1d whose training statistics are averaged
over ranks with one all-gather of [mean, mean-of-squares] (collective C2/C3 of SURVEY.md 2.3); plain BN1d in eval
or when world size is 1 (norm.py:57-58)."""
import torch
from torch import distributed as dist
from torch import nn
from torch.autograd.function import Function

from ..compat.registry import NORM_LAYERS


class AllReduce(Function):
    """sum over ranks via all_gather in forward, all_reduce of the gradient in backward (norm.py:9-24)."""

    @staticmethod
    def forward(ctx, x):
        parts = [torch.zeros_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, x, async_op=False)
        return torch.stack(parts, dim=0).sum(dim=0)

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        dist.all_reduce(grad, async_op=False)
        return grad


@NORM_LAYERS.register_module("naiveSyncBN1dCustom")
class NaiveSyncBatchNorm1dCustom(nn.BatchNorm1d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.fp16_enabled = False

    def forward(self, x):
        assert x.dtype == torch.float32, f"input should be in float32 type, got {x.dtype}"
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or not self.training:
            return super().forward(x)
        assert x.shape[0] > 0, "SyncBN does not support empty inputs"
        two_d = x.dim() == 2
        if two_d:
            x = x.unsqueeze(2)
        C = x.shape[1]
        stats = torch.cat([x.mean(dim=[0, 2]), (x * x).mean(dim=[0, 2])], dim=0)
        stats = AllReduce.apply(stats) * (1.0 / dist.get_world_size())
        mean, meansqr = torch.split(stats, C)
        var = meansqr - mean * mean
        with torch.no_grad():
            self.running_mean += self.momentum * (mean.detach() - self.running_mean)
            self.running_var += self.momentum * (var.detach() - self.running_var)
        scale = self.weight * torch.rsqrt(var + self.eps)
        shift = self.bias - mean * scale
        out = x * scale.reshape(1, -1, 1) + shift.reshape(1, -1, 1)
        return out.squeeze(2) if two_d else out
