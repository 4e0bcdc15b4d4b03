"""mmdet `ResNet` stand-in for the r50 / r101 image-backbone configs (configs/nus/srfdet_*_r50_nusc_LC.py:`img_backbone`).

Same constructor arguments the configs use and the same state_dict names as mmdet's ResNet (`conv1`, `bn1`,
`layer{1-4}.{i}.conv{1-3}/bn{1-3}`, `layer*.0.downsample.{0,1}`), so `Pretrained` checkpoints load by key.  Plain dense
convolutions on MIOpen; in inference every conv -> BN -> ReLU goes through dense.conv_bn_act (one fused BN + ReLU pass).
Stages with `dcn=dict(type='DCNv2')` (only configs/others/srfdet_dvoxel_waymo_LC.py) use compat/dcn.py for the 3x3
convolution of their bottlenecks.
"""
import torch
from torch import nn

from ..dense import conv_bn_act
from .cnn import BaseModule, build_norm_layer
from .registry import BACKBONES

_ARCH = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)), 50: ("bottleneck", (3, 4, 6, 3)),
         101: ("bottleneck", (3, 4, 23, 3)), 152: ("bottleneck", (3, 8, 36, 3))}


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, style, norm_cfg, dcn=None):
        super().__init__()
        s1, s2 = (1, stride) if style == "pytorch" else (stride, 1)   # caffe style strides the 1x1, pytorch the 3x3
        self.conv1 = nn.Conv2d(inplanes, planes, 1, s1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        if dcn is not None:
            from .dcn import ModulatedDeformConv2dPack
            self.conv2 = ModulatedDeformConv2dPack(planes, planes, 3, s2, 1, deform_groups=dcn.get("deform_groups", 1), bias=False)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, s2, 1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = build_norm_layer(norm_cfg, planes * 4)[1]
        self.downsample = downsample

    def forward(self, x):
        out = conv_bn_act(self.conv1, self.bn1, True, x)
        out = conv_bn_act(self.conv2, self.bn2, True, out)
        out = conv_bn_act(self.conv3, self.bn3, False, out)
        idt = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], False, x)
        return torch.relu_(out + idt)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, style, norm_cfg, dcn=None):
        super().__init__()
        assert dcn is None, "DCN is defined for bottleneck blocks only (as in mmdet)"
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.downsample = downsample

    def forward(self, x):
        out = conv_bn_act(self.conv1, self.bn1, True, x)
        out = conv_bn_act(self.conv2, self.bn2, False, out)
        idt = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], False, x)
        return torch.relu_(out + idt)


@BACKBONES.register_module()
class ResNet(BaseModule):
    def __init__(self, depth, in_channels=3, stem_channels=64, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch", deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True,
                 pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        if depth not in _ARCH:
            raise KeyError(f"invalid depth {depth} for ResNet")
        if dcn is not None:
            dcn = dict(dcn)
            if dcn.pop("type", "DCNv2") != "DCNv2" or dcn.pop("fallback_on_stride", False):
                raise NotImplementedError("only dcn=dict(type='DCNv2', fallback_on_stride=False) is provided")
        if deep_stem or avg_down or plugins is not None or any(d != 1 for d in dilations):
            raise NotImplementedError("only the plain ResNet variants the SRFDet3D configs use are provided")
        kind, blocks = _ARCH[depth]
        block = _Bottleneck if kind == "bottleneck" else _BasicBlock
        self.out_indices, self.frozen_stages, self.norm_eval = tuple(out_indices), frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, 2, 3, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, stem_channels)[1]
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        inplanes = stem_channels
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            layers = []
            for j in range(blocks[i]):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                                         build_norm_layer(norm_cfg, planes * block.expansion)[1])
                layers.append(block(inplanes, planes, stride, down, style, norm_cfg, dcn if (dcn is not None and stage_with_dcn[i]) else None))
                inplanes = planes * block.expansion
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.maxpool(conv_bn_act(self.conv1, self.bn1, True, x))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self
