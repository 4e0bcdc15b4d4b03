"""Inference-time execution of the dense conv -> BatchNorm2d -> ReLU chains of the path (SECONDCustom, FPN, VoVNet).

The convolutions stay on MIOpen; the eval-mode BatchNorm2d and the ReLU behind each of them run as ONE in-place pass
(`ops.channel_affine`, csrc/dense.hip) instead of two kernels.  Module structure, parameter names and the training /
autograd / autocast behaviour are untouched: the fused route is taken only for fp32 CUDA tensors with grad disabled and
BatchNorm in eval mode, everything else goes through the modules as written.
"""
import torch
from torch import nn

from . import ops


def _fold_bn2d(bn):
    """scale = gamma / sqrt(var + eps), shift = beta - mean * scale, cached until a BN tensor changes."""
    vers = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
            bn.weight.data_ptr(), bn.running_var.data_ptr())
    cache = getattr(bn, "_srf_fold", None)
    if cache is None or cache[0] != vers:
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
        cache = (vers, scale.contiguous(), shift.contiguous())
        bn._srf_fold = cache
    return cache[1], cache[2]


def _foldable(bn):
    return isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.track_running_stats and bn.affine


def fusable(x):
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not torch.is_autocast_enabled()


def bn_act_(y, bn, relu):
    """In place on the contiguous NCHW conv output y."""
    scale, shift = _fold_bn2d(bn)
    return ops.channel_affine(y, scale, shift, relu, out=y)


def conv_bn_act(conv, bn, relu, x):
    y = conv(x)
    if _foldable(bn) and fusable(y) and y.is_contiguous():
        return bn_act_(y, bn, relu)
    y = bn(y)
    return torch.relu_(y) if relu else y


def run_sequential(seq, x):
    """nn.Sequential forward with every [Conv2d, BatchNorm2d(eval), (ReLU)] run through `conv_bn_act`."""
    mods = list(seq.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        if (isinstance(m, nn.Conv2d) and i + 1 < len(mods) and _foldable(mods[i + 1]) and fusable(x)):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = conv_bn_act(m, mods[i + 1], relu, x)
            i += 3 if relu else 2
        else:
            x = m(x)
            i += 1
    return x
