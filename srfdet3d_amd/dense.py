"""Inference-time execution of the dense conv -> BatchNorm2d -> ReLU chains of the path (SECONDCustom, FPN, VoVNet).

The convolutions stay on MIOpen; the eval-mode BatchNorm2d and the ReLU behind each of them run as ONE in-place pass
(`ops.channel_affine`, csrc/dense.hip) instead of two kernels.  Module structure, parameter names and the training /
autograd / autocast behaviour are untouched: the fused route is taken only for fp32 CUDA tensors with grad disabled and
BatchNorm in eval mode, everything else goes through the modules as written.
"""
import os

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


def _is_dw3x3s2(conv, x):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == conv.in_channels == conv.out_channels and conv.bias is None
            and x.dim() == 4 and x.is_contiguous())


def conv_bn_act(conv, bn, relu, x):
    if _is_dw3x3s2(conv, x) and _foldable(bn) and fusable(x):
        # the depthwise stair of the proposal generator: convolution, BatchNorm and ReLU in one streaming kernel
        scale, shift = _fold_bn2d(bn)
        return ops.dwconv3x3s2(x, conv.weight, scale, shift, relu)
    from . import train_conv
    if _train_fusable(x):
        y = train_conv.conv_bn_act(conv, bn, relu, x)   # training: conv + eval BatchNorm + ReLU as one autograd node
        if y is not None:
            return y
    y = train_conv.conv2d(conv, x)   # training: the 3x3 layers on srf_wino43 (forward and data gradient), else conv(x)
    if _foldable(bn) and fusable(y) and y.is_contiguous() and y.shape[0] * y.shape[1] <= 65535:  # grid.y of the kernel
        return bn_act_(y, bn, relu)
    y = train_conv.bn_eval(bn, y)    # eval-mode BatchNorm under autograd as the affine map it is; else bn(y)
    return torch.relu_(y) if relu else y


def _packed_1x1(conv):
    vers = (conv.weight._version, conv.weight.data_ptr())
    cache = getattr(conv, "_srf_packed", None)
    if cache is None or cache[0] != vers:
        cache = (vers, ops.pack_conv1x1_weights(conv.weight.detach()))
        conv._srf_packed = cache
    return cache[1]


def _is_plain_1x1(conv):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.dilation == (1, 1))


def conv1x1_cat_bn_act(conv, bn, relu, xs):
    """conv1x1(cat(xs, 1)) -> BN(eval) -> ReLU.  On the fused route the concatenation is never built: srf_conv1x1 reads
    the sources in place and applies the folded BatchNorm (or the conv bias) and the ReLU as its epilogue."""
    if (_is_plain_1x1(conv) and (bn is None or _foldable(bn)) and fusable(xs[0]) and ops.conv1x1_supported(xs, conv.out_channels)
            and sum(x.shape[1] for x in xs) == conv.in_channels):
        if bn is not None:
            scale, shift = _fold_bn2d(bn)
            if conv.bias is not None:
                shift = shift + conv.bias * scale
        else:
            scale, shift = None, conv.bias
        return ops.conv1x1(xs, _packed_1x1(conv), conv.out_channels, scale, shift, relu)
    x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
    if bn is None:
        from . import train_conv
        y = train_conv.conv2d(conv, x)
        return torch.relu_(y) if relu else y
    return conv_bn_act(conv, bn, relu, x)


def _train_fusable(x):
    """under autograd: conv -> eval-mode BatchNorm -> ReLU may run as train_conv._ConvAffineRelu (conv_bn_act decides per layer)"""
    return (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and os.environ.get("SRF_TRAIN_FUSED", "1") != "0")


def run_sequential(seq, x):
    """nn.Sequential forward with every [Conv2d, BatchNorm2d(eval), (ReLU)] run through `conv_bn_act`."""
    mods = list(seq.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        if (isinstance(m, nn.Conv2d) and i + 1 < len(mods) and _foldable(mods[i + 1]) and (fusable(x) or _train_fusable(x))):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = conv_bn_act(m, mods[i + 1], relu, x)
            i += 3 if relu else 2
        elif isinstance(m, nn.Conv2d):
            from . import train_conv
            x = train_conv.conv2d(m, x)
            i += 1
        elif isinstance(m, nn.BatchNorm2d):
            from . import train_conv
            x = train_conv.bn_eval(m, x)
            i += 1
        else:
            x = m(x)
            i += 1
    return x


def linear_graph_safe(lin, x, relu=False, act=None):
    """nn.Linear for inference on the GPU through this library's f32-MFMA GEMM (ops.linear) instead of rocBLAS / hipBLASLt.

    Why: for skinny problems (the DPG layers have M = batch size) the BLAS libraries pick split-K kernels that accumulate
    with atomics into an output they clear first; captured into a hipGraph such a layer returned correct values on the first
    replay and different ones from the second on (found on `dpg_fc1_img`, 900 -> 1500 at M = 1; round 1 blamed the memset node,
    tools/micro/graph_memset.hip shows memset nodes do take effect -- the cause inside the library is not established).
    ops.linear is one ordered fma chain per output and clears nothing.  K is zero-padded to a multiple of 4 (cached) when needed; an
    `x` that already has the padded width (ops.nhwc_pool_sum writes it so) is taken as is.  relu: ReLU in the GEMM's epilogue (the
    module `act` is applied instead where the layer runs on torch)."""
    if not (fusable(x) and x.dim() == 2 and lin.weight.is_cuda):
        y = lin(x[:, :lin.in_features])
        return (act(y) if act is not None else torch.relu(y)) if relu else y
    K = lin.in_features
    w = lin.weight
    if K % 4:
        vers = (w._version, w.data_ptr())
        cache = getattr(lin, "_srf_padded", None)
        if cache is None or cache[0] != vers:
            Kp = (K + 3) // 4 * 4
            wp = w.new_zeros((w.shape[0], Kp))
            wp[:, :K] = w.detach()
            cache = (vers, wp)
            lin._srf_padded = cache
        w = cache[1]
        if x.shape[1] != w.shape[1]:
            x = torch.nn.functional.pad(x, (0, w.shape[1] - K))
    return ops.linear(x.contiguous(), w, lin.bias, relu1=bool(relu))
