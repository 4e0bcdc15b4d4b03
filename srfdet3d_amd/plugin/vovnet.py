"""`VoVNet` image backbone (mmdet3d_plugin/models/backbones/vovnet.py:269-374): one-shot-aggregation stages with eSE
attention; V-99-eSE feeds the LC configs (configs/nus/srfdet_voxel_nusc_LC.py:44-54).

The modules own the parameters; fp32 inference on the GPU executes them channels-last on the hand-written kernels
(`nhwc.vovnet_forward`: srf_stem_conv_nchw, srf_wino43 / srf_wino3x3, srf_conv_gemm_nhwc, srf_conv1x1_nhwc*, the streaming
kernels of csrc/nhwc.hip -- SURVEY.md 8f-2, 90 % of the LC frame); training and the optional autocast mode go through the
modules as written (torch -> MIOpen, `train_conv.py` for the trainable 3x3 layers).  Module and parameter names follow the
reference so its checkpoints load
(`stem.stem_1/conv.weight`, `stage3.OSA3_2.layers.0.OSA3_2_0/conv.weight`, `...concat.OSA3_2_concat/conv.weight`,
`...ese.fc.weight`).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..compat.cnn import BaseModule
from ..compat.registry import BACKBONES
from .. import ops
from ..dense import conv1x1_cat_bn_act, fusable, run_sequential

# name: (stem widths, per-stage conv width, per-stage output width, layers per block, blocks per stage, depthwise)
SPECS = {
    "V-19-slim-dw-eSE": ((64, 64, 64), (64, 80, 96, 112), (112, 256, 384, 512), 3, (1, 1, 1, 1), True),
    "V-19-dw-eSE": ((64, 64, 64), (128, 160, 192, 224), (256, 512, 768, 1024), 3, (1, 1, 1, 1), True),
    "V-19-slim-eSE": ((64, 64, 128), (64, 80, 96, 112), (112, 256, 384, 512), 3, (1, 1, 1, 1), False),
    "V-19-eSE": ((64, 64, 128), (128, 160, 192, 224), (256, 512, 768, 1024), 3, (1, 1, 1, 1), False),
    "V-39-eSE": ((64, 64, 128), (128, 160, 192, 224), (256, 512, 768, 1024), 5, (1, 1, 2, 2), False),
    "V-57-eSE": ((64, 64, 128), (128, 160, 192, 224), (256, 512, 768, 1024), 5, (1, 1, 4, 3), False),
    "V-99-eSE": ((64, 64, 128), (128, 160, 192, 224), (256, 512, 768, 1024), 5, (1, 3, 9, 3), False),
}


def _cbr(cin, cout, name, k, stride=1):
    """conv(k x k, no bias) -> BN -> ReLU with the reference's '<name>/conv|norm|relu' child names."""
    return [(f"{name}/conv", nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)),
            (f"{name}/norm", nn.BatchNorm2d(cout)), (f"{name}/relu", nn.ReLU(inplace=True))]


def _dw(cin, cout, name, stride=1):
    return [(f"{name}/dw_conv3x3", nn.Conv2d(cin, cout, 3, stride, 1, groups=cout, bias=False)),
            (f"{name}/pw_conv1x1", nn.Conv2d(cin, cout, 1, bias=False)),
            (f"{name}/pw_norm", nn.BatchNorm2d(cout)), (f"{name}/pw_relu", nn.ReLU(inplace=True))]


class eSEModule(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.fc = nn.Conv2d(channel, channel, kernel_size=1)

    def forward(self, x, identity=None):
        from .. import train_conv
        if train_conv.ese_eligible(self, x, identity):
            return train_conv.ese_apply(self, x, identity)     # training on channels-last tensors: one autograd node
        if fusable(x) and x.shape[0] <= 8 and x.shape[1] % 4 == 0:
            # the 1x1 conv on the pooled (N, C, 1, 1) tensor is a GEMV: fc + bias + hard sigmoid in one launch
            gate = ops.ese_gate(x.mean(dim=(2, 3)), self.fc.weight, self.fc.bias)
        else:
            gate = F.relu6(self.fc(x.mean(dim=(2, 3), keepdim=True)) + 3.0) / 6.0
        if (fusable(x) and x.is_contiguous() and (identity is None or identity.is_contiguous())
                and x.shape[0] * x.shape[1] <= 65535):
            # gate multiply (+ the OSA identity add) in one in-place pass
            return ops.channel_affine(x, gate.reshape(-1), None, False, out=x, residual=identity)
        out = x * gate
        return out if identity is None else out + identity


class OSAModule(nn.Module):
    def __init__(self, cin, width, cout, n_layers, name, identity=False, depthwise=False):
        super().__init__()
        self.identity = identity
        self.reduce = None
        c = cin
        if depthwise and cin != width:
            self.conv_reduction = nn.Sequential(OrderedDict(_cbr(cin, width, f"{name}_reduction_0", 1)))
            self.reduce = self.conv_reduction
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            seq = _dw(width, width, f"{name}_{i}") if depthwise else _cbr(c, width, f"{name}_{i}", 3)
            self.layers.append(nn.Sequential(OrderedDict(seq)))
            c = width
        self.concat = nn.Sequential(OrderedDict(_cbr(cin + n_layers * width, cout, f"{name}_concat", 1)))
        self.ese = eSEModule(cout)

    def forward(self, x):
        from .. import train_conv
        if train_conv.enabled() and train_conv.osa_eligible(self, x):
            # training behind the frozen prefix: layers + concat convolution as one autograd node over one channels-last buffer
            return self.ese(train_conv.osa_chain(self, x), x if self.identity else None)
        feats = [x]
        y = run_sequential(self.reduce, x) if self.reduce is not None else x
        for layer in self.layers:
            y = run_sequential(layer, y)
            feats.append(y)
        conv, bn = self.concat[0], self.concat[1]
        out = conv1x1_cat_bn_act(conv, bn, True, feats)
        return self.ese(out, x if self.identity else None)


def _stage(cin, width, cout, n_blocks, n_layers, idx, depthwise):
    mods = OrderedDict()
    if idx != 2:
        mods["Pooling"] = nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True)
    mods[f"OSA{idx}_1"] = OSAModule(cin, width, cout, n_layers, f"OSA{idx}_1", depthwise=depthwise)
    for i in range(n_blocks - 1):
        name = f"OSA{idx}_{i + 2}"
        mods[name] = OSAModule(cout, width, cout, n_layers, name, identity=True, depthwise=depthwise)
    return nn.Sequential(mods)


@BACKBONES.register_module()
class VoVNet(BaseModule):
    def __init__(self, spec_name, input_ch=3, out_features=None, frozen_stages=-1, norm_eval=True, pretrained=None,
                 init_cfg=None):
        super().__init__(init_cfg)
        self.frozen_stages = frozen_stages
        self.norm_eval = norm_eval
        stem_w, widths, outs, n_layers, n_blocks, dw = SPECS[spec_name]
        self._out_features = list(out_features)
        two = _dw if dw else (lambda a, b, n, s=1: _cbr(a, b, n, 3, s))
        self.stem = nn.Sequential(OrderedDict(_cbr(input_ch, stem_w[0], "stem_1", 3, 2) + two(stem_w[0], stem_w[1], "stem_2", 1)
                                              + two(stem_w[1], stem_w[2], "stem_3", 2)))
        ins = (stem_w[2],) + tuple(outs[:-1])
        self.stage_names = []
        for i in range(4):
            name = f"stage{i + 2}"
            self.stage_names.append(name)
            self.add_module(name, _stage(ins[i], widths[i], outs[i], n_blocks[i], n_layers, i + 2, dw))

    def forward(self, x):
        from .. import nhwc
        if nhwc.enabled() and nhwc.vovnet_supported(self, x):
            # fp32 inference on the GPU: channels-last execution on the Winograd / GEMM kernels of csrc/conv.hip
            return nhwc.vovnet_forward(self, x)
        out = OrderedDict()
        done = self._frozen_prefix(x, out)   # training: the frozen stages on the inference kernels, without autograd
        if done is None:
            x = run_sequential(self.stem, x)
            if "stem" in self._out_features:
                out["stem"] = x
        else:
            x = done
        for name in self.stage_names:
            if done is not None and int(name[5:]) <= self.frozen_stages + 1:
                continue
            for m in getattr(self, name).children():
                if (isinstance(m, nn.MaxPool2d) and fusable(x) and m.kernel_size == 3 and m.stride == 2 and m.padding == 0
                        and m.ceil_mode and m.dilation == 1):
                    x = ops.maxpool3s2_ceil(x)  # torch's kernel also tracks indices: 2.4 TB/s on these maps
                else:
                    x = m(x)
            if name in self._out_features:
                out[name] = x
        return out

    def _frozen_prefix(self, x, out):
        """With `frozen_stages` >= 1 the stem and stage2 .. stage{frozen_stages + 1} carry no gradient and their BatchNorms are in
        eval mode even while the rest trains (vovnet.py `_freeze_stages`, config `frozen_stages=2, norm_eval=True`): run
        them channels-last on the Winograd / GEMM kernels under no_grad, as inference does, and hand NCHW copies to the
        trainable remainder (module path, autograd).  Returns the last frozen stage's output or None (not applicable)."""
        from .. import nhwc
        if not (torch.is_grad_enabled() and self.frozen_stages >= 1 and nhwc.enabled() and not x.requires_grad):
            return None
        last = f"stage{self.frozen_stages + 1}"
        if last not in self.stage_names:
            return None
        frozen = [self.stem] + [getattr(self, f"stage{i + 1}") for i in range(1, self.frozen_stages + 1)]
        if any(p.requires_grad for m in frozen for p in m.parameters()) or any(m.training for m in frozen):
            return None
        with torch.no_grad():
            if not nhwc.vovnet_supported(self, x):
                return None
            part, cur = nhwc.vovnet_forward(self, x, upto=last)
            from .. import train_conv
            keep_cl = train_conv.enabled()       # the trainable remainder runs channels-last (train_conv.py): no copy at all
            for k, v in part.items():
                out[k] = v if keep_cl else v.contiguous()   # else NCHW for the module path on MIOpen
            if last in out:
                return out[last]
            return nhwc.nchw_view(cur) if keep_cl else nhwc.nchw_view(cur).contiguous()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for m in [self.stem] + [getattr(self, f"stage{i + 1}") for i in range(1, self.frozen_stages + 1)]:
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self
