"""Channels-last (NHWC) inference of the camera branch on the hand-written kernels of csrc/conv.hip and csrc/nhwc.hip:
VoVNet (vovnet.py:269-374) -> FPN (configs/nus/srfdet_voxel_nusc_LC.py:55-64) -> `img_convs` (srfdet_head.py:404-416).

The torch modules stay the owners of the parameters (state_dict names untouched); this file only EXECUTES them:

* every 3x3 / stride 1 convolution runs on `srf_wino3x3` (Winograd F(2x2, 3x3) on the f32 MFMA) with the eval BatchNorm
  (or the bias) and the ReLU as its epilogue;
* an OSA block owns ONE pixel-major buffer of Cin + 5 w channels: the block input sits in slice 0, each 3x3 branch writes
  its slice, and the 1x1 `concat` convolution (`srf_conv1x1_nhwc`) reads the buffer as a plain matrix -- the
  torch.cat of vovnet.py:222 is never built;
* eSE: pixel mean (epilogue of the concat convolution, `srf_conv1x1_nhwc_pooled`) -> fc + hard sigmoid (`srf_ese_gate`) -> gate multiply + identity add in one pass
  that writes straight into slice 0 of the next block's buffer (`srf_nhwc_affine`);
* the stride-2 stem layers: stem_1 (3 -> 64) is a streaming kernel from the NCHW images to channels-last
  (`srf_stem_conv_nchw`), stem_3 an implicit-im2col GEMM (`srf_conv_gemm_nhwc`); nothing of the branch runs on MIOpen.

Tensors handed to the rest of the model are logical NCHW views with channels_last strides, so every consumer that only
looks at shapes keeps working and the RoI gather finds its channels-last operand without a copy.

Taken only for fp32 CUDA inference with BatchNorm in eval mode (`dense.fusable` / `dense._foldable`); anything else goes
through the modules as written.
"""
import os

import torch
from torch import nn

from . import ops
from .dense import _fold_bn2d, _foldable, fusable


def enabled():
    """SRF_IMG_NHWC=0 sends the camera branch through torch / MIOpen as in round 1 (A/B switch for tests and benchmarks)."""
    import os
    return os.environ.get("SRF_IMG_NHWC", "1") != "0"


def nhwc_view(x):
    """Logical (N, C, H, W) tensor with channels_last strides -> its (N, H, W, C) view."""
    return x.permute(0, 2, 3, 1)


def nchw_view(x):
    return x.permute(0, 3, 1, 2)


def is_channels_last(x):
    return x.dim() == 4 and x.stride(1) == 1 and x.shape[1] > 1


def _cached(mod, key, vers, make):
    cache = getattr(mod, key, None)
    if cache is None or cache[0] != vers:
        cache = (vers, make())
        setattr(mod, key, cache)
    return cache[1]


_CACHE_KEYS = ("_srf_wino", "_srf_wino43", "_srf_gemm", "_srf_gemm_direct", "_srf_gemm_split", "_srf_cgemm", "_srf_cgemm_split", "_srf_packed")


def invalidate_caches(model):
    """Drops the packed-weight images cached on the convolution modules of `model`.  The caches are keyed on (weight._version,
    data_ptr): optimiser steps, `copy_` and `load_state_dict` bump the version, but an in-place update THROUGH `.data`
    (`p.data.mul_()`, old-style EMA) does not -- call this after such an update.  The detector calls it from `train()` and
    after `load_state_dict`."""
    for m in model.modules():
        for k in _CACHE_KEYS:
            if hasattr(m, k):
                delattr(m, k)


def _wino_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_wino", (w._version, w.data_ptr()), lambda: ops.pack_wino3x3_weights(w.detach()))


def _gemm_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_gemm", (w._version, w.data_ptr()), lambda: ops.pack_conv1x1_nhwc_weights(w.detach()))


def _is_conv(conv, k, stride=1):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (k, k) and conv.stride == (stride, stride)
            and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1) and conv.groups == 1)


def _affine_of(conv, bn):
    """(scale, shift) of the layer's epilogue: folded eval BatchNorm (and bias), or the bias alone."""
    if bn is not None:
        scale, shift = _fold_bn2d(bn)
        if conv.bias is not None:
            shift = shift + conv.bias * scale
        return scale, shift
    return None, conv.bias


def _wino43_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_wino43", (w._version, w.data_ptr()), lambda: ops.pack_wino43_weights(w.detach()))


def wino43_enabled():
    """SRF_WINO43=0 keeps every 3x3 layer on Winograd F(2x2, 3x3) (A/B switch for tests and benchmarks)."""
    return os.environ.get("SRF_WINO43", "1") != "0"


def use_wino43(x, cout, out=None):
    """F(4x4, 3x3) (`srf_wino43`: direct FLOPs / 4, plus an HBM pass that writes the transformed input) where it is the faster
    of the two kernels: every layer from 96 input channels up.  Below that (VoVNet stem_2: 64 -> 64 on 464 x 800) the
    transform pass costs more than the saved MFMA time (tools/micro/wino43_bench.hip: 954 against 928 us)."""
    return wino43_enabled() and x.shape[3] >= 96 and ops.wino43_supported(x, cout, out)


def conv3x3(x, conv, bn=None, relu=False, out=None):
    """x: NHWC slice; conv: nn.Conv2d 3x3 / stride 1 / padding 1."""
    scale, shift = _affine_of(conv, bn)
    if use_wino43(x, conv.out_channels, out):
        return ops.wino43(x, _wino43_weights(conv), conv.out_channels, scale, shift, relu, out=out)
    return ops.wino3x3(x, _wino_weights(conv), conv.out_channels, scale, shift, relu, out=out)


def _gemm_direct_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_gemm_direct", (w._version, w.data_ptr()), lambda: ops.pack_conv1x1_nhwc_direct_weights(w.detach()))


def _gemm_split_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_gemm_split", (w._version, w.data_ptr()), lambda: ops.pack_conv1x1_nhwc_split_weights(w.detach()))


def conv1x1(x, conv, bn=None, relu=False, out=None, pool=False, top=None):
    scale, shift = _affine_of(conv, bn)
    # the operand orders are packed lazily: a layer only ever packs the one its launch selects
    return ops.conv1x1_nhwc(x, lambda: _gemm_weights(conv), conv.out_channels, scale, shift, relu, out=out, pool=pool, top=top,
                            packed_direct=lambda: _gemm_direct_weights(conv), packed_split=lambda: _gemm_split_weights(conv))


def wino_ok(conv, cin):
    return _is_conv(conv, 3) and cin % 8 == 0


def _img_fits(H, W, ld):
    """One image of an (N, H, W, ld) f32 buffer inside the 2^30-byte per-image range of srf_wino3x3 (ops.wino3x3_supported)."""
    return 4 * H * W * ld < (1 << 30)


def gemm_ok(conv, cin):
    return _is_conv(conv, 1) and cin % 32 == 0


# ---- VoVNet ----------------------------------------------------------------------------------------------------------
def _cbr(seq):
    """[conv, bn, relu] children of an nn.Sequential built by vovnet._cbr; None if it has another form."""
    mods = list(seq.children())
    if len(mods) == 3 and isinstance(mods[0], nn.Conv2d) and _foldable(mods[1]) and isinstance(mods[2], nn.ReLU):
        return mods[0], mods[1]
    return None


def vovnet_supported(net, x):
    from .plugin.vovnet import OSAModule
    if not (fusable(x) and x.dim() == 4):
        return False
    stem = list(net.stem.children())
    if len(stem) != 9:
        return False
    for i in (0, 3, 6):
        if not (isinstance(stem[i], nn.Conv2d) and _foldable(stem[i + 1]) and isinstance(stem[i + 2], nn.ReLU)):
            return False
    if not (wino_ok(stem[3], stem[3].in_channels) and stem[3].bias is None):
        return False
    c0, c6 = stem[0], stem[6]
    if not (c0.kernel_size == (3, 3) and c0.stride == (2, 2) and c0.padding == (1, 1) and c0.groups == 1 and c0.in_channels <= 4
            and c0.out_channels == 64 and c6.kernel_size == (3, 3) and c6.stride == (2, 2) and c6.padding == (1, 1)
            and strided_ok(c6, c6.in_channels)):
        return False
    # every buffer the executor makes must fit the kernels' per-image 32-bit offsets (ops.wino3x3_supported: 4 H W ld < 2^30)
    H, W = (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1          # after stem_1
    if not _img_fits(H, W, 64):
        return False
    H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1                              # after stem_3
    for name in net.stage_names:
        for m in getattr(net, name).children():
            if isinstance(m, nn.MaxPool2d):
                H, W = ops.pool3s2_out(H), ops.pool3s2_out(W)
            elif isinstance(m, OSAModule):
                convs = [_cbr(layer) for layer in m.layers]
                if any(c is None for c in convs):
                    return False
                if not _img_fits(H, W, convs[0][0].in_channels + sum(c[0].out_channels for c in convs)):
                    return False
    for name in net.stage_names:
        for m in getattr(net, name).children():
            if isinstance(m, nn.MaxPool2d):
                if not (m.kernel_size == 3 and m.stride == 2 and m.padding == 0 and m.ceil_mode and m.dilation == 1):
                    return False
            elif isinstance(m, OSAModule):
                if m.reduce is not None:
                    return False
                cin = None
                for layer in m.layers:
                    cb = _cbr(layer)
                    if cb is None or not wino_ok(cb[0], cb[0].in_channels):
                        return False
                    cin = cin or cb[0].in_channels
                cc = _cbr(m.concat)
                if cc is None or not gemm_ok(cc[0], cc[0].in_channels) or cc[0].out_channels % 4 or cc[0].out_channels > 1024:
                    return False
            else:
                return False
    return True


def _osa_forward(m, buf, cin, dst):
    """One OSA block.  buf: (N, H, W, cin + L w) with the block input in [..., :cin]; dst: NHWC slice that receives the
    block output (gate * concat (+ input))."""
    off = cin
    src = buf[..., :cin]
    for layer in m.layers:
        conv, bn = _cbr(layer)
        w = conv.out_channels
        out = buf[..., off:off + w]
        conv3x3(src, conv, bn, True, out=out)
        src, off = out, off + w
    conv, bn = _cbr(m.concat)
    t, mean = conv1x1(buf, conv, bn, True, pool=True)     # eSE average pool from the convolution's own epilogue
    res = buf[..., :cin] if m.identity else None
    if ops.ese_apply_supported(t, t.shape[3]):
        ops.ese_apply(t, mean, m.ese.fc.weight, m.ese.fc.bias, residual=res, out=dst)   # gate GEMV + multiply (+ identity): one launch
    else:
        gate = ops.ese_gate(mean, m.ese.fc.weight, m.ese.fc.bias)
        ops.nhwc_affine(t, scale=gate, residual=res, out=dst)
    return dst


def vovnet_forward(net, x, upto=None):
    """x (N, 3, H, W) f32 -> OrderedDict of the requested stage outputs (logical NCHW, channels_last strides).
    upto = a stage name: stop after that stage and return (outputs so far, that stage's NHWC output) -- the frozen prefix
    of the backbone during training (`VoVNet.forward`)."""
    from collections import OrderedDict
    from .plugin.vovnet import OSAModule
    out = OrderedDict()
    stem = list(net.stem.children())
    # stem_1 (3 -> 64, stride 2): NCHW images -> channels-last, BatchNorm + ReLU in the same kernel
    s1, b1 = _affine_of(stem[0], stem[1])
    y = ops.stem_conv_nchw(x.contiguous(), stem[0].weight, s1, b1, True)
    # stem_2 (64 -> 64) Winograd; stem_3 (64 -> 128, stride 2) implicit-im2col GEMM, written straight into the first block's buffer
    y = conv3x3(y, stem[3], stem[4], True)
    cur = None          # finished NHWC tensor (stage output) when not already inside a block buffer
    pending = y         # stem_2 output: stem_3 runs when the first buffer exists
    if "stem" in net._out_features:
        cur = conv_strided(y, stem[6], stem[7], True)
        pending = None
        out["stem"] = nchw_view(cur)
    for name in net.stage_names:
        mods = list(getattr(net, name).children())
        blocks = [m for m in mods if isinstance(m, OSAModule)]
        pool = any(isinstance(m, nn.MaxPool2d) for m in mods)
        first = blocks[0]
        cin = _cbr(first.layers[0])[0].in_channels
        width = sum(_cbr(l)[0].out_channels for l in first.layers)
        src = pending if pending is not None else cur
        N, H, W, _ = src.shape
        if pending is not None:
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1   # stem_3: 3x3, stride 2, padding 1
        elif pool:
            H, W = ops.pool3s2_out(H), ops.pool3s2_out(W)
        buf = torch.empty((N, H, W, cin + width), dtype=torch.float32, device=src.device)
        if pending is not None:
            conv_strided(pending, stem[6], stem[7], True, out=buf[..., :cin])
            pending = None
        elif pool:
            ops.nhwc_maxpool3s2_ceil(src, out=buf[..., :cin])
        else:
            buf[..., :cin].copy_(src)
        for i, m in enumerate(blocks):
            cout = _cbr(m.concat)[0].out_channels
            if i + 1 < len(blocks):
                nxt = blocks[i + 1]
                ncin = _cbr(nxt.layers[0])[0].in_channels
                nwidth = sum(_cbr(l)[0].out_channels for l in nxt.layers)
                nbuf = torch.empty((N, H, W, ncin + nwidth), dtype=torch.float32, device=src.device)
                _osa_forward(m, buf, cin, nbuf[..., :cout])
                buf, cin = nbuf, ncin
            else:
                cur = torch.empty((N, H, W, cout), dtype=torch.float32, device=src.device)
                _osa_forward(m, buf, cin, cur)
        if name in net._out_features:
            out[name] = nchw_view(cur)
        if upto is not None and name == upto:
            return out, cur
    return out


def _strided_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_cgemm", (w._version, w.data_ptr()), lambda: ops.pack_conv_gemm_weights(w))


def _strided_split_weights(conv):
    w = conv.weight
    return _cached(conv, "_srf_cgemm_split", (w._version, w.data_ptr()), lambda: ops.pack_conv_gemm_split_weights(w))


def strided_ok(conv, cin):
    return (isinstance(conv, nn.Conv2d) and conv.groups == 1 and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and cin % 32 == 0)


def conv_strided(x, conv, bn=None, relu=False, out=None):
    """A convolution the Winograd kernel does not cover (stride 2) as an implicit-im2col GEMM on the f32 MFMA
    (`srf_conv_gemm_nhwc`): deterministic, where MIOpen's channels-last choice is an atomic split-K kernel."""
    scale, shift = _affine_of(conv, bn)
    return ops.conv_gemm_nhwc(x, lambda: _strided_weights(conv), conv.out_channels, conv.kernel_size, conv.stride[0], conv.padding[0],
                              scale, shift, relu, out=out, packed_split=lambda: _strided_split_weights(conv))


def to_nhwc(x):
    """Contiguous NCHW f32 tensor -> NHWC tensor (LDS tile transpose)."""
    return nhwc_view(ops.to_channels_last(x))


# ---- SECONDCustom ----------------------------------------------------------------------------------------------------
def second_supported(net, x):
    if not (fusable(x) and x.dim() == 4 and x.shape[1] % 8 == 0):
        return False
    widest = max([x.shape[1]] + [m.out_channels for st in net.blocks for m in st.children() if isinstance(m, nn.Conv2d)])
    if not _img_fits(x.shape[2], x.shape[3], widest):
        return False
    for stage in net.blocks:
        mods = list(stage.children())
        if len(mods) % 3:
            return False
        for j in range(0, len(mods), 3):
            conv, bn, act = mods[j:j + 3]
            if not (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.groups == 1
                    and conv.dilation == (1, 1) and _foldable(bn) and isinstance(act, nn.ReLU)):
                return False
            if conv.stride == (1, 1) and conv.in_channels % 8:
                return False
            if conv.stride != (1, 1) and not strided_ok(conv, conv.in_channels):
                return False
    return True


def second_forward(net, x):
    """SECONDCustom.forward (second_custom.py:78-91) on channels-last maps: 3x3 / stride 1 layers on srf_wino3x3, the
    stride-2 heads of the later blocks on srf_conv_gemm_nhwc."""
    y = nhwc_view(x) if is_channels_last(x) else to_nhwc(x)
    outs = []
    for stage in net.blocks:
        mods = list(stage.children())
        for j in range(0, len(mods), 3):
            conv, bn = mods[j], mods[j + 1]
            y = conv3x3(y, conv, bn, True) if conv.stride == (1, 1) else conv_strided(y, conv, bn, True)
        outs.append(nchw_view(y))
    return tuple(outs)


# ---- FPN -------------------------------------------------------------------------------------------------------------
def fpn_supported(fpn, inputs):
    n = len(fpn.lateral_convs)
    if fpn.start_level != 0 or fpn.backbone_end_level != fpn.num_ins or len(inputs) != n:
        return False
    if fpn.num_outs != n and not (fpn.num_outs > n and fpn.add_extra_convs == "on_output"):
        return False
    for cm in list(fpn.fpn_convs)[n:]:   # extra stride-2 convolutions: srf_conv_gemm_nhwc
        if cm.with_norm and not _foldable(getattr(cm, cm.norm_name)):
            return False
        if not strided_ok(cm.conv, cm.conv.in_channels):
            return False
        if cm.with_activation and not isinstance(cm.activate, nn.ReLU):
            return False
    if fpn.upsample_cfg.get("mode", "nearest") != "nearest" or len(fpn.upsample_cfg) != 1:
        return False
    for x, lat, fc in zip(inputs, fpn.lateral_convs, list(fpn.fpn_convs)[:n]):
        if not (fusable(x) and is_channels_last(x)):
            return False
        try:   # a channels-last view that is not a channel slice of a pixel-major buffer (or too large) goes to the module path
            ld = ops.nhwc_ld(nhwc_view(x))
        except RuntimeError:
            return False
        if ld % 4 or x.data_ptr() % 16 or not _img_fits(x.shape[2], x.shape[3], max(ld, lat.conv.out_channels)):
            return False
        for cm in (lat, fc):
            if cm.with_activation and not isinstance(cm.activate, nn.ReLU):
                return False
            if cm.with_norm and not _foldable(getattr(cm, cm.norm_name)):
                return False
        if not (gemm_ok(lat.conv, lat.conv.in_channels) and wino_ok(fc.conv, fc.conv.in_channels)):
            return False
        if lat.conv.out_channels % 4:
            return False
    return True


def _cm(cm, x, fn):
    bn = getattr(cm, cm.norm_name) if cm.with_norm else None
    return fn(x, cm.conv, bn, cm.with_activation)


class ConsumedLevels(list):
    """Pyramid levels a per-level consumer (the head's `img_convs`) has already been applied to (see `level_consumer`)."""


class level_consumer:
    """Within this context the channels-last forward of THIS FPN instance hands every finished level to `fn(i, x_nhwc) ->
    y_nhwc` as part of that level's chain (lateral -> output convolution -> consumer).  graphs.GraphedImageBranch uses it to
    run the head's `img_convs` (srfdet_head.py:404-416) inside the camera graph, where the chains of the coarse levels are
    captured as parallel branches: their few-workgroup kernels (29 x 50 and 58 x 100 maps cover 65 % / 80 % of the CUs) run
    beside the lateral GEMMs and the finest level's kernels instead of after them.  The consumer is an attribute of the neck it
    is meant for (no module-global state: another FPN's forward is not affected); fn = None is a no-op context."""

    def __init__(self, fpn, fn):
        self.fpn, self.fn = fpn, fn

    def __enter__(self):
        if self.fn is not None:
            self.fpn._srf_level_consumer = self.fn
        return self

    def __exit__(self, *exc):
        if self.fn is not None:
            self.fpn._srf_level_consumer = None
        return False


_CHAIN_STREAMS = []


def _chain_stream(i):
    while len(_CHAIN_STREAMS) <= i:
        _CHAIN_STREAMS.append(torch.cuda.Stream())
    return _CHAIN_STREAMS[i]


def fpn_forward(fpn, inputs):
    # laterals from the top level down: the top-down step (`laterals[i - 1] += upsample(laterals[i])`) is the epilogue of the
    # lateral convolution of level i - 1 (srf_conv1x1_nhwc_topdown): no separate pass over the finer map
    n = len(fpn.lateral_convs)
    convs = list(fpn.fpn_convs)
    consumer = getattr(fpn, "_srf_level_consumer", None)
    if len(convs) > n:
        consumer = None   # extra levels hang off the last output: one chain
    # under a graph capture the chain of every coarse level forks off as soon as its lateral is final
    fork_from = int(os.environ.get("SRF_FPN_FORK", "2"))   # first level whose chain forks; 0: none
    fork = consumer is not None and inputs[0].is_cuda and torch.cuda.is_current_stream_capturing() and fork_from > 0
    main = torch.cuda.current_stream() if fork else None

    def chain(i):
        o = _cm(convs[i], lats[i], conv3x3)
        return consumer(i, o) if consumer is not None else o

    lats = [None] * n
    outs = [None] * n
    used = []
    for i in range(n - 1, -1, -1):
        cm = fpn.lateral_convs[i]
        bn = getattr(cm, cm.norm_name) if cm.with_norm else None
        lats[i] = conv1x1(nhwc_view(inputs[i]), cm.conv, bn, cm.with_activation, top=lats[i + 1] if i + 1 < n else None)
        if fork and i >= fork_from:
            s = _chain_stream(i)
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs[i] = chain(i)
            used.append(s)
    for i in range(n):
        if outs[i] is None:
            outs[i] = chain(i)
    for s in used:
        main.wait_stream(s)   # the join; `lats` and `outs` stay referenced until here, so no block is reused across streams
    for cm in convs[n:]:   # add_extra_convs='on_output': stride-2 3x3 on the previous output
        src = outs[-1]
        if fpn.relu_before_extra_convs and len(outs) > n:
            src = torch.relu(src)
        outs.append(_cm(cm, src, conv_strided))
    res = tuple(nchw_view(o) for o in outs)
    return ConsumedLevels(res) if consumer is not None else res
