"""Training-time 3x3 convolutions of the camera branch on the library's Winograd kernel (config 4 of BASELINE.json trains
VoVNet stages 4-5, the image FPN, `img_convs` and the head: tools/train.py:220-234, vovnet.py:354-374).

torch's autograd runs these layers on MIOpen: its fp32 3x3 algorithm is the VALU Winograd F(2x3) kernel for the forward AND the
data gradient (142 launches, 68 of the 340 ms of kernel time of a bs = 2 step: profiles/r03_train_step_kernels_before.md), wrapped
in NCHW <-> NHWC transposes for the weight gradient.  Here

* forward        y = conv(x, W) + b              -> `srf_wino43` (F(4x4, 3x3) on the f32 MFMA) on the channels-last tensor;
* data gradient  dx = conv(dy, rot180(W)^T)      -> the SAME kernel: a stride-1 3x3 convolution of dy with the weights rotated
                                                    by 180 degrees and their in / out channel axes swapped (packed once per step);
* weight grad    dW = sum_p dy[p] x[p + tap]     -> aten.convolution_backward with only the weight mask set (MIOpen's NHWC
                                                    implicit-GEMM kernels; both operands are already channels-last, so the
                                                    transposes around them disappear);
* bias grad      db = sum_p dy[p].

Tensors stay logical NCHW with channels_last strides, which every torch op of the module path preserves (BatchNorm in eval
mode -- `norm_eval=True` -- ReLU, cat, max_pool2d, nearest interpolate), so nothing is copied between the layers.
SRF_TRAIN_CONV=0 switches back to torch's convolution (A/B switch for tests and benchmarks).
"""
import os
import weakref

import torch
from torch import nn

from . import ops


_DEBUG = [] if os.environ.get("SRF_TRAIN_CONV_DEBUG") else None   # developer: (input shape, Cout, events) of every weight-gradient call


def debug_report():
    torch.cuda.synchronize()
    rows = {}
    for shp, co, e0, e1 in _DEBUG or []:
        r = rows.setdefault((shp, co), [0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
    return sorted(((k, n, ms / n) for k, (n, ms) in rows.items()), key=lambda t: -t[2] * t[1])


def enabled():
    return os.environ.get("SRF_TRAIN_CONV", "1") != "0"


def _weight_grad(gn, xn, weight, k):
    """dW of a stride-1 convolution from the channels-last output gradient gn (N, H, W, Cout) and input xn (N, H, W, Cin).
    `srf_conv_wgrad_nhwc` (round 5: an f32 GEMM over the pixels on the bf16 MFMA, exact three-way split of both operands, pixel ranges
    added in a fixed order -- deterministic) where it applies; else (SRF_TRAIN_WGRAD=0, maps narrower than 32 pixels, tensors of 2 GB)
    the library route of rounds 3-4: aten.convolution_backward = MIOpen's float-atomic split-K kernels for 3x3, one rocBLAS GEMM for 1x1."""
    Co, Ci = weight.shape[0], weight.shape[1]
    cols = k * k * Ci if (k > 1 and Ci % 32 == 0) else None            # 3x3 layers: the column tiles run over the flattened (tap, channel) axis
    fill = ((Co * cols) / float(((Co + 127) // 128) * ((cols + 127) // 128) * 128 * 128) if cols is not None
            else (Co * Ci) / float(((Co + 127) // 128) * ((Ci + 127) // 128) * 128 * 128))   # share of the kernel's 128 x 128 tiles that is real
    # (192 -> 192, VoVNet stage 4, with every tap's channels in tiles of their own: 56 % -- 543 us against MIOpen's 465; flattened: 72 %;
    # every other trainable shape of config 4 is 1.1-3x faster on the kernel: profiles/r05_wgrad_bench.txt)
    want = os.environ.get("SRF_TRAIN_WGRAD", "1")
    if want != "0" and (fill >= 0.6 or k == 1 or want == "2") and ops.conv_wgrad_supported(gn, xn, k):
        return ops.conv_wgrad_nhwc(gn, xn, k)
    Cout, Cin = weight.shape[0], weight.shape[1]
    if k == 1:
        return (gn.reshape(-1, Cout).t() @ xn.reshape(-1, Cin)).view(Cout, Cin, 1, 1)
    return torch.ops.aten.convolution_backward(gn.permute(0, 3, 1, 2), xn.permute(0, 3, 1, 2), weight, None, (1, 1), (1, 1), (1, 1), False,
                                               (0, 0), 1, (False, True, False))[1]


def eligible(conv, x):
    """A trainable (or gradient-carrying) 3x3 / stride 1 / padding 1 convolution on an fp32 GPU tensor under autograd."""
    return (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and not torch.is_autocast_enabled() and type(conv) is nn.Conv2d and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0 and conv.in_channels >= 32
            and (x.requires_grad or conv.weight.requires_grad) and x.shape[0] * x.shape[2] * x.shape[3] * max(conv.in_channels, conv.out_channels) * 4 < (1 << 32) - 16)


def _nhwc(t):
    """logical NCHW tensor -> its (N, H, W, C) view over channels-last storage (a copy only if it was not channels-last)."""
    return t.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


class _Wino43Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        xn = _nhwc(x)
        Cout = weight.shape[0]
        y = ops.wino43(xn, ops.pack_wino43_weights(weight.detach()), Cout, None, None if bias is None else bias.detach(), False)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.keep_cl = x.stride(1) == 1
        y = y.permute(0, 3, 1, 2)             # logical NCHW, channels-last strides
        # The caller's memory format is kept: a channels-last input (the camera branch behind the frozen prefix, whose BatchNorms
        # are in eval mode and run as affine maps) gets a channels-last output without a copy; an NCHW-contiguous input (the BEV
        # backbone of the LiDAR-only configs, BatchNorm in TRAIN mode) gets NCHW back -- MIOpen's training batch-norm on a
        # channels-last tensor crashed the process behind a stride-2 convolution of the BEV FPN
        # (profiles/r03_fault_pytest_segfault_20261005.log).
        return y if ctx.keep_cl else y.contiguous()

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gyn = _nhwc(gy)
        gy_cl = gyn.permute(0, 3, 1, 2)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            w_t = weight.detach().flip(2, 3).transpose(0, 1).contiguous()          # (Cin, Cout, 3, 3), rotated by 180 degrees
            gx = ops.wino43(gyn, ops.pack_wino43_weights(w_t), weight.shape[1]).permute(0, 3, 1, 2)
            if not ctx.keep_cl:
                gx = gx.contiguous()
        if ctx.needs_input_grad[1]:
            dbg = _DEBUG is not None
            if dbg:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            gw = _weight_grad(gyn, _nhwc(x), weight, 3)
            if dbg:
                e1.record()
                _DEBUG.append((tuple(x.shape), weight.shape[0], e0, e1))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy_cl.sum(dim=(0, 2, 3))
        return gx, gw, gb


class _ConvAffineRelu(torch.autograd.Function):
    """conv (3x3 / stride 1 on `srf_wino43`, or 1x1 on the library's GEMM) -> BatchNorm2d in EVAL mode (`norm_eval=True`: an affine
    map with trainable gamma / beta) -> ReLU as ONE forward launch -- the inference kernel with the folded BatchNorm and the ReLU in
    its epilogue -- and a backward pass that starts with ONE streaming kernel (`srf_nhwc_affine_relu_bwd`: ReLU mask, scale, and
    the two column sums gamma / beta need).  As torch ops the same chain is conv, multiply, add, relu_ forward and
    threshold_backward, multiply and two strided column sums backward: eight passes over each layer's output instead of two.
    Only y is saved (not the convolution's own output z): sum gu z = (sum gu y - t sum gu) / s since z = (y - t) / s wherever the mask
    lets a gradient through -- which needs s away from 0: `conv_bn_act` takes this node only for layers whose min |s| >= 1e-3 max |s|
    (`gamma_well_conditioned`; anything else runs under plain autograd), and a frozen gamma needs no d gamma at all."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, mean, var, eps, relu):
        xn = _nhwc(x)
        Cout, Cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
        fold = ops.bn_eval_fold(gamma.detach(), beta.detach(), mean, var, eps)     # s, t0, inv in one launch
        s, t0 = fold[0], fold[1]
        t = t0 if bias is None else t0 + bias.detach() * s
        w = weight.detach()
        if k == 3:
            yn = ops.wino43(xn, ops.pack_wino43_weights(w), Cout, s, t, relu)
        else:
            w2 = w.reshape(Cout, Cin)
            yn = ops.conv1x1_nhwc(xn, lambda: ops.pack_conv1x1_nhwc_weights(w2), Cout, s, t, relu,
                                  packed_split=lambda: ops.pack_conv1x1_nhwc_split_weights(w2))
        ctx.save_for_backward(x, weight, yn, fold, mean)
        ctx.relu, ctx.has_bias, ctx.k = bool(relu), bias is not None, k
        return yn.permute(0, 3, 1, 2)          # logical NCHW, channels-last strides

    @staticmethod
    def backward(ctx, gy):
        x, weight, yn, fold, mean = ctx.saved_tensors
        s = fold[0]
        Cout, Cin = weight.shape[0], weight.shape[1]
        gz, sums = ops.nhwc_affine_relu_bwd(_nhwc(gy), yn, s, ctx.relu)
        need = ctx.needs_input_grad
        gx = gw = gb = ggamma = gbeta = None
        if need[3] or need[4]:
            # u = (z - mean) inv gamma + beta:  d gamma = inv (sum gu z - mean sum gu),  sum gu z = (sum gu y - t0 sum gu) / s;  d beta = sum gu
            pg = ops.bn_eval_grads(sums, fold, mean)
            ggamma = pg[0] if need[3] else None
            gbeta = pg[1] if need[4] else None
        if ctx.has_bias and need[2]:
            gb = sums[0] * s
        w = weight.detach()
        if need[0]:
            if ctx.k == 3:
                w_t = w.flip(2, 3).transpose(0, 1).contiguous()           # (Cin, Cout, 3, 3), rotated by 180 degrees
                gx = ops.wino43(gz, ops.pack_wino43_weights(w_t), Cin).permute(0, 3, 1, 2)
            else:
                w_t = w.reshape(Cout, Cin).t().contiguous()              # (Cin, Cout): dx = gz W
                gx = ops.conv1x1_nhwc(gz, lambda: ops.pack_conv1x1_nhwc_weights(w_t), Cin,
                                      packed_split=lambda: ops.pack_conv1x1_nhwc_split_weights(w_t)).permute(0, 3, 1, 2)
        if need[1]:
            gw = _weight_grad(gz, _nhwc(x), weight, ctx.k)
        return gx, gw, gb, ggamma, gbeta, None, None, None, None


def fused_eligible(conv, bn, x):
    """conv -> eval-mode BatchNorm2d (-> ReLU) under autograd on a channels-last tensor, in the shapes `_ConvAffineRelu` covers."""
    if not (isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.track_running_stats and bn.affine and x.dim() == 4 and x.is_cuda
            and x.stride(1) == 1 and conv.out_channels % 4 == 0 and conv.out_channels <= 1024):
        return False
    if eligible(conv, x):
        return ops.wino43_supported(x.permute(0, 2, 3, 1), conv.out_channels)
    return (eligible_1x1(conv, x) and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] * max(conv.in_channels, conv.out_channels) * 512 < (1 << 31) * 128)


GAMMA_GUARD_RATIO = 1e-3   # min |s| / max |s| of a layer below which `_ConvAffineRelu` is not used (s = gamma / sqrt(var + eps))
_GUARD = weakref.WeakKeyDictionary()   # BatchNorm2d -> ((gamma version, gamma pointer, var version), well conditioned?)


def _guard_key(bn):
    return (bn.weight._version, bn.weight.data_ptr(), bn.running_var._version)


def gamma_well_conditioned(bn):
    """`_ConvAffineRelu` rebuilds sum gu z from the saved OUTPUT, (sum gu y - t sum gu) / s: a channel with s = 0 has no such
    term at all (its true d gamma is not zero, so a zero-initialised or pruned channel could never recover) and a small |s|
    amplifies the rounding of two column sums by 1 / s.  The node is therefore used only for layers whose
    min |s| >= GAMMA_GUARD_RATIO max |s|; any other layer runs conv2d / bn_eval / relu under plain autograd, which is exact.
    The verdict is cached per (gamma version, pointer); when one is stale, the verdicts of ALL layers seen so far are refreshed
    in one batch -- two concatenations, two segment reductions and ONE device -> host copy per optimiser step, not one per layer."""
    ent = _GUARD.get(bn)
    if ent is not None and ent[0] == _guard_key(bn):
        return ent[1]
    dev = bn.weight.device
    mods = [m for m, e in list(_GUARD.items()) if m is not bn and m.weight.device == dev and e[0] != _guard_key(m)] + [bn]
    with torch.no_grad():
        lengths = [m.weight.numel() for m in mods]
        g = torch.cat([m.weight.detach().reshape(-1) for m in mods]).abs()
        v = torch.cat([m.running_var.reshape(-1) for m in mods])
        eps = torch.repeat_interleave(torch.tensor([float(m.eps) for m in mods], dtype=v.dtype), torch.tensor(lengths)).to(dev)
        s = g * torch.rsqrt(v + eps)
        lt = torch.tensor(lengths, device=dev)
        lo = torch.segment_reduce(s, "min", lengths=lt)
        hi = torch.segment_reduce(s, "max", lengths=lt)
        ok = ((hi > 0) & (lo >= GAMMA_GUARD_RATIO * hi) & torch.isfinite(hi)).tolist()     # the one synchronisation
    for m, o in zip(mods, ok):
        _GUARD[m] = (_guard_key(m), bool(o))
    return _GUARD[bn][1]


def conv_bn_act(conv, bn, relu, x):
    """The fused training route when it applies, else None (the caller then runs conv2d / bn_eval / relu)."""
    if fused_eligible(conv, bn, x) and (not bn.weight.requires_grad or gamma_well_conditioned(bn)):
        return _ConvAffineRelu.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bool(relu))
    return None


class _OSAChain(torch.autograd.Function):
    """The body of a VoVNet OSA block (vovnet.py:208-230) -- L x [conv 3x3 -> eval BatchNorm -> ReLU] in a chain, then the 1x1 `concat`
    convolution over [x, y_0 .. y_{L-1}] -> eval BatchNorm -> ReLU -- as ONE autograd node over ONE channels-last buffer
    cat = [x | y_0 | .. | y_{L-1}]:

    * forward: every layer reads its input slice of `cat` and writes its output slice (the kernels take a pixel pitch), the concat
      convolution reads `cat` in place: no torch.cat, and `cat` is all that is saved (it holds every layer's input AND output);
    * backward: the concat convolution's data gradient g_cat has the same layout; layer i's output gradient is g_cat's slice PLUS the
      data gradient of layer i + 1 -- added inside the ReLU / BatchNorm backward pass (`srf_nhwc_affine_relu_bwd2`), read from the slice
      in place.  As separate `_ConvAffineRelu` nodes autograd made that sum with a strided add per layer (213 launches, 7.3 ms per
      step) after a contiguous copy of the slice (4.3 ms) and torch.cat had copied every output once more forward (2.9 ms):
      profiles/r05_train_step_kernels.md.
    Same arithmetic per layer as `_ConvAffineRelu` (same kernels, same d gamma / d beta formulas); the one difference is where the two
    gradients of an output are added (one f32 add either way)."""

    @staticmethod
    def forward(ctx, x, eps, *params):
        L = len(params) // 5 - 1
        xn = _nhwc(x)
        N, H, W, Cin = xn.shape
        ws = [params[5 * i] for i in range(L + 1)]
        widths = [int(w.shape[0]) for w in ws[:L]]
        Ctot = Cin + sum(widths)
        cat = torch.empty((N, H, W, Ctot), dtype=torch.float32, device=x.device)
        cat[..., :Cin].copy_(xn)
        aff = []
        lo, hi = 0, Cin            # the input slice of the next layer
        for i in range(L):
            w, gamma, beta, mean, var = params[5 * i:5 * i + 5]
            fold = ops.bn_eval_fold(gamma.detach(), beta.detach(), mean, var, eps[i])
            ops.wino43(cat[..., lo:hi], ops.pack_wino43_weights(w.detach()), widths[i], fold[0], fold[1], True, out=cat[..., hi:hi + widths[i]])
            aff.append(fold)
            lo, hi = hi, hi + widths[i]
            if i == 0:
                lo = Cin
        wc, gamma, beta, mean, var = params[5 * L:5 * L + 5]
        Cout = int(wc.shape[0])
        fold = ops.bn_eval_fold(gamma.detach(), beta.detach(), mean, var, eps[L])
        w2 = wc.detach().reshape(Cout, Ctot)
        yc = ops.conv1x1_nhwc(cat, lambda: ops.pack_conv1x1_nhwc_weights(w2), Cout, fold[0], fold[1], True,
                              packed_split=lambda: ops.pack_conv1x1_nhwc_split_weights(w2))
        aff.append(fold)
        ctx.save_for_backward(cat, yc, *ws, *[params[5 * i + 3] for i in range(L + 1)], *aff)
        ctx.L, ctx.Cin, ctx.widths = L, Cin, widths
        return yc.permute(0, 3, 1, 2)          # logical NCHW, channels-last strides

    @staticmethod
    def backward(ctx, gy):
        L, Cin, widths = ctx.L, ctx.Cin, ctx.widths
        sv = ctx.saved_tensors
        cat, yc = sv[0], sv[1]
        ws, means, aff = sv[2:3 + L], sv[3 + L:4 + 2 * L], sv[4 + 2 * L:]      # aff[i] = (3, C): s, t0, inv of layer i
        need = ctx.needs_input_grad            # x, eps, then 5 per layer: weight, gamma, beta, mean, var
        grads = [None] * (2 + 5 * (L + 1))

        def affine_grads(i, sums):
            if need[2 + 5 * i + 1] or need[2 + 5 * i + 2]:
                pg = ops.bn_eval_grads(sums, aff[i], means[i])
                if need[2 + 5 * i + 1]:
                    grads[2 + 5 * i + 1] = pg[0]
                if need[2 + 5 * i + 2]:
                    grads[2 + 5 * i + 2] = pg[1]

        # the concat convolution
        Ctot = cat.shape[3]
        wc = ws[L].detach()
        Cout = wc.shape[0]
        gz, sums = ops.nhwc_affine_relu_bwd(_nhwc(gy), yc, aff[L][0], True)
        affine_grads(L, sums)
        w_t = wc.reshape(Cout, Ctot).t().contiguous()
        g_cat = ops.conv1x1_nhwc(gz, lambda: ops.pack_conv1x1_nhwc_weights(w_t), Ctot,
                                 packed_split=lambda: ops.pack_conv1x1_nhwc_split_weights(w_t))
        if need[2 + 5 * L]:
            grads[2 + 5 * L] = _weight_grad(gz, cat, ws[L], 1)
        del gz
        # the chain, last layer first
        carry = None
        out_hi = Ctot
        for i in range(L - 1, -1, -1):
            out_lo = out_hi - widths[i]
            in_lo, in_hi = (0, Cin) if i == 0 else (out_lo - widths[i - 1], out_lo)
            gz, sums = ops.nhwc_affine_relu_bwd(g_cat[..., out_lo:out_hi], cat[..., out_lo:out_hi], aff[i][0], True, gy2=carry)
            affine_grads(i, sums)
            w = ws[i].detach()
            if need[2 + 5 * i]:
                grads[2 + 5 * i] = _weight_grad(gz, cat[..., in_lo:in_hi], ws[i], 3)
            carry = None
            if i > 0 or need[0]:
                w_r = w.flip(2, 3).transpose(0, 1).contiguous()           # (Cin, Cout, 3, 3), rotated by 180 degrees
                carry = ops.wino43(gz, ops.pack_wino43_weights(w_r), in_hi - in_lo)
            out_hi = out_lo
        if need[0]:
            grads[0] = carry.add_(g_cat[..., :Cin]).permute(0, 3, 1, 2)
        return tuple(grads)


class _ESEApply(torch.autograd.Function):
    """VoVNet's eSE module applied to an OSA block's output (vovnet.py:165-177, :225-228): y = out * hsigmoid(fc(mean(out))) (+ identity)
    as one autograd node on channels-last tensors.  Forward: pixel mean, the gate GEMV, one streaming pass (the inference kernels).
    Backward: ONE pass for the pixel sums of g * out (what the gate's gradient needs), the (N, C)-sized arithmetic of the gate, ONE pass
    d out = g * gate + d mean / HW; the identity's gradient is g itself.  As torch ops the same backward is two multiplies, a
    reduction, an expand and an add over the block's output -- five passes over 53 M elements per stage-4 block."""

    @staticmethod
    def forward(ctx, out, identity, weight, bias):
        xn = _nhwc(out)
        N, H, W, C = xn.shape
        mean = ops.nhwc_colmean(xn)
        gate = ops.ese_gate(mean, weight.detach(), bias.detach())
        y = ops.nhwc_affine(xn, scale=gate, residual=None if identity is None else _nhwc(identity))
        ctx.save_for_backward(out, mean, gate, weight, bias)
        ctx.has_identity = identity is not None
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        out, mean, gate, weight, bias = ctx.saved_tensors
        gn, xn = _nhwc(g), _nhwc(out)
        N, H, W, C = xn.shape
        need = ctx.needs_input_grad
        d_out = d_w = d_b = None
        w2 = weight.detach().reshape(C, C)
        s = ops.nhwc_colsum_prod(gn, xn)                                   # (N, C): sum_p g * out = d gate
        pre = torch.addmm(bias.detach(), mean, w2.t())                     # fc(mean): the gate's argument
        dpre = s * ((pre > -3.0) & (pre < 3.0)).to(s.dtype) / 6.0          # hsigmoid = relu6(. + 3) / 6
        if need[2]:
            d_w = (dpre.t() @ mean).view_as(weight)
        if need[3]:
            d_b = dpre.sum(0)
        if need[0]:
            dmean = (dpre @ w2) / float(H * W)                            # back through the pixel mean
            d_out = ops.nhwc_affine(gn, scale=gate, shift=dmean).permute(0, 3, 1, 2)
        return d_out, (g if (ctx.has_identity and need[1]) else None), d_w, d_b


def ese_eligible(mod, x, identity):
    """The eSE module under autograd on a channels-last f32 GPU tensor (SRF_TRAIN_ESE=0: the torch ops)."""
    C = x.shape[1] if x.dim() == 4 else 0
    return (enabled() and os.environ.get("SRF_TRAIN_ESE", "1") != "0" and torch.is_grad_enabled() and x.dim() == 4 and x.is_cuda
            and x.dtype == torch.float32 and x.stride(1) == 1 and not torch.is_autocast_enabled() and C % 4 == 0 and 0 < C <= 1024
            and x.shape[0] <= 65535 and (x.requires_grad or mod.fc.weight.requires_grad)
            and (identity is None or (tuple(identity.shape) == tuple(x.shape) and identity.stride(1) == 1))
            and type(mod.fc) is nn.Conv2d and mod.fc.kernel_size == (1, 1) and mod.fc.bias is not None)


def ese_apply(mod, x, identity):
    return _ESEApply.apply(x, identity, mod.fc.weight, mod.fc.bias)


def osa_eligible(block, x):
    """An OSA block whose body can run as `_OSAChain`: plain 3x3 layers (no reduction / depthwise form), every layer and the concat
    convolution fit `_ConvAffineRelu`'s conditions (bias-free convolutions, eval-mode BatchNorms with well-conditioned gammas), and
    the input is channels-last under autograd.  SRF_TRAIN_OSA=0 keeps the per-layer nodes."""
    if os.environ.get("SRF_TRAIN_OSA", "1") == "0" or os.environ.get("SRF_TRAIN_FUSED", "1") == "0" or block.reduce is not None:
        return False
    if not (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.stride(1) == 1 and len(block.layers) >= 1):
        return False
    if torch.is_autocast_enabled():
        return False
    N, _, H, W = x.shape
    seqs = list(block.layers) + [block.concat]
    cin = x.shape[1]
    ctot = cin
    for j, seq in enumerate(seqs):
        mods = list(seq.children())
        if len(mods) != 3 or type(mods[0]) is not nn.Conv2d or not isinstance(mods[2], nn.ReLU) or mods[0].bias is not None:
            return False
        conv, bn = mods[0], mods[1]
        if not (isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.track_running_stats and bn.affine):
            return False
        if conv.dilation != (1, 1) or conv.groups != 1 or conv.stride != (1, 1) or conv.padding_mode != "zeros":
            return False
        last = j == len(seqs) - 1
        if not last:
            if not (conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.in_channels == cin and cin % 8 == 0 and cin >= 32
                    and conv.out_channels % 8 == 0 and conv.out_channels <= 1024):
                return False
            cin = conv.out_channels
            ctot += cin
        elif not (conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.in_channels == ctot and ctot % 32 == 0
                  and conv.out_channels % 32 == 0 and conv.out_channels <= 1024 and H * W > 1):
            return False
        if bn.weight.requires_grad and not gamma_well_conditioned(bn):
            return False
    if not (x.requires_grad or any(p.requires_grad for seq in seqs for p in seq.parameters())):
        return False
    if N * H * W * ctot * 4 >= (1 << 31) or N * H * W * max(ctot, seqs[-1][0].out_channels) * 512 >= (1 << 31) * 128:
        return False
    return N * ((H + 3) // 4) * ((W + 3) // 4) < (1 << 31) - 64   # (the Winograd kernel's tile count)


def osa_chain(block, x):
    seqs = list(block.layers) + [block.concat]
    params, eps = [], []
    for seq in seqs:
        conv, bn = seq[0], seq[1]
        params += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        eps.append(float(bn.eps))
    return _OSAChain.apply(x, tuple(eps), *params)


class _DepthwiseNative(torch.autograd.Function):
    """Depthwise convolutions (the stride-2 stair of the proposal generator, srfdet_head.py:520-537) on torch's own depthwise
    kernels in BOTH directions.  The forward already kept MIOpen away (its choice is a naive reference kernel: 34 ms on the
    finest image level); under autograd the backward ran outside that context and MIOpen's weight gradient for the grouped
    convolution was one 59 ms CK batched-GEMM launch per step (profiles/r03_train_step_kernels_before.md)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, groups):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, groups, bias is not None)
        with torch.backends.cudnn.flags(enabled=False):
            return torch.nn.functional.conv2d(x, weight, bias, stride, padding, 1, groups)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, groups, has_bias = ctx.cfg
        with torch.backends.cudnn.flags(enabled=False):
            gx, gw, gb = torch.ops.aten.convolution_backward(gy.contiguous(), x, weight, [weight.shape[0]] if has_bias else None, stride, padding,
                                                             (1, 1), False, (0, 0), groups,
                                                             (ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]))
        return gx, gw, gb, None, None, None


def eligible_depthwise(conv, x):
    return (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(conv) is nn.Conv2d
            and conv.groups > 1 and conv.groups == conv.in_channels == conv.out_channels and conv.dilation == (1, 1)
            and conv.padding_mode == "zeros" and (x.requires_grad or conv.weight.requires_grad))


def eligible_1x1(conv, x):
    # only for tensors that ARE channels-last already: an NCHW pipeline keeps torch's convolution (and its memory format)
    return (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.stride(1) == 1
            and not torch.is_autocast_enabled() and type(conv) is nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and x.shape[2] * x.shape[3] > 1
            and (x.requires_grad or conv.weight.requires_grad))


def conv2d(conv, x):
    """conv(x) with the training-time 3x3 layers routed through `_Wino43Conv` and the 1x1 layers through a plain GEMM on the
    (pixels, channels) view of the channels-last tensor (forward, data and weight gradient are then three rocBLAS GEMMs under
    torch's own autograd; MIOpen's channels-last choices for 1x1 layers cost 24 ms forward and one 59 ms weight-gradient launch
    per step: profiles/r03_train_step_kernels_before.md).  Every other case is the module itself."""
    if eligible(conv, x):
        return _Wino43Conv.apply(x, conv.weight, conv.bias)
    if eligible_depthwise(conv, x):
        return _DepthwiseNative.apply(x.contiguous(), conv.weight, conv.bias, conv.stride, conv.padding, conv.groups)
    if eligible_1x1(conv, x):
        if (os.environ.get("SRF_TRAIN_CONV1X1", "1") != "0" and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0
                and conv.out_channels <= 1024 and x.shape[0] * x.shape[2] * x.shape[3] * max(conv.in_channels, conv.out_channels) * 512 < (1 << 31) * 128):
            return _Conv1x1.apply(x, conv.weight, conv.bias)
        xn = _nhwc(x)
        N, H, W, C = xn.shape
        y = torch.nn.functional.linear(xn.reshape(N * H * W, C), conv.weight.view(conv.out_channels, C), conv.bias)
        return y.view(N, H, W, conv.out_channels).permute(0, 3, 1, 2)   # channels-last in, channels-last out
    return conv(x)


class _Conv1x1(torch.autograd.Function):
    """A bias-carrying 1x1 convolution without a BatchNorm behind it (the image FPN's lateral convolutions) on the library's GEMMs in all
    three directions: forward and data gradient on `srf_conv1x1_nhwc` (the split GEMM from 128 tiles up), the weight gradient on
    `srf_conv_wgrad_nhwc` (deterministic; as torch's `linear` it was one rocBLAS TN GEMM over 1.1 M pixels at 62 TFLOP/s)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xn = _nhwc(x)
        Cout, Cin = weight.shape[0], weight.shape[1]
        w2 = weight.detach().reshape(Cout, Cin)
        yn = ops.conv1x1_nhwc(xn, lambda: ops.pack_conv1x1_nhwc_weights(w2), Cout, None, None if bias is None else bias.detach(), False,
                              packed_split=lambda: ops.pack_conv1x1_nhwc_split_weights(w2))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return yn.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        Cout, Cin = weight.shape[0], weight.shape[1]
        gn = _nhwc(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            w_t = weight.detach().reshape(Cout, Cin).t().contiguous()
            gx = ops.conv1x1_nhwc(gn, lambda: ops.pack_conv1x1_nhwc_weights(w_t), Cin,
                                  packed_split=lambda: ops.pack_conv1x1_nhwc_split_weights(w_t)).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(gn, _nhwc(x), weight, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gn.sum(dim=(0, 1, 2))
        return gx, gw, gb


def bn_eval(bn, y):
    """BatchNorm2d in eval mode under autograd (`norm_eval=True`: vovnet.py:371) as the affine map it is, y * s + t with
    s = gamma / sqrt(var + eps), t = beta - mean * s: the gradients of gamma and beta come out of two column sums instead of
    torch's channels-last batch-norm backward (0.36 ms per layer, 26 ms per step).  Anything else: the module."""
    if (enabled() and torch.is_grad_enabled() and y.is_cuda and y.dtype == torch.float32 and isinstance(bn, nn.BatchNorm2d)
            and not bn.training and bn.track_running_stats and bn.affine and not torch.is_autocast_enabled()):
        s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        t = bn.bias - bn.running_mean * s
        return y * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    return bn(bn_train_input(bn, y))


def bn_train_input(bn, y):
    """What a BatchNorm in TRAINING mode may be handed.  MIOpen's training batch-norm dies with a host segmentation fault on a
    channels-last tensor of batch size 1 with a 23 x 23 map -- (1, 128, 23, 23) with strides (67712, 1, 2944, 128), also (1, 64, 23, 23),
    forward alone included; 24 x 24 and 46 x 46 at batch size 1, the same tensor at batch size 2, the same shape NCHW-contiguous and eval
    mode are all fine (tools/bn_channels_last_probe.py, every case in its own process; profiles/r04_bn_channels_last_probe.txt).  That was the round-3 crash: a bs = 1 LiDAR-only training step whose BEV FPN extras
    (stride-2 Conv2d -> BatchNorm2d in train mode) received channels-last tensors from `_Wino43Conv`.  Guarded here, at the one
    place every module-path BatchNorm goes through: a non-contiguous 4-D GPU tensor is made NCHW-contiguous before a train-mode
    BatchNorm sees it (any batch size: the trigger is a library bug, not something to steer around case by case).  Eval-mode
    BatchNorms (the camera branch's `norm_eval=True`) never get here with a copy."""
    if (isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training and y.is_cuda and y.dim() == 4 and not y.is_contiguous()):
        return y.contiguous()
    return y
