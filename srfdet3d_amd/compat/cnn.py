"""mmcv.cnn stand-ins: build_conv_layer / build_norm_layer / build_activation_layer / ConvModule, and
mmcv.runner's BaseModule (state_dict names match mmcv: ConvModule.{conv,bn,activate})."""
import torch
from torch import nn

from .registry import ACTIVATION_LAYERS, CONV_LAYERS, NORM_LAYERS

CONV_LAYERS.register_module("Conv2d", module=nn.Conv2d)
CONV_LAYERS.register_module("Conv1d", module=nn.Conv1d)
CONV_LAYERS.register_module("Conv", module=nn.Conv2d)
NORM_LAYERS.register_module("BN", module=nn.BatchNorm2d)
NORM_LAYERS.register_module("BN1d", module=nn.BatchNorm1d)
NORM_LAYERS.register_module("BN2d", module=nn.BatchNorm2d)
NORM_LAYERS.register_module("LN", module=nn.LayerNorm)
ACTIVATION_LAYERS.register_module("ReLU", module=nn.ReLU)
ACTIVATION_LAYERS.register_module("Tanh", module=nn.Tanh)
ACTIVATION_LAYERS.register_module("Sigmoid", module=nn.Sigmoid)


class BaseModule(nn.Module):
    """mmcv.runner.BaseModule: nn.Module that remembers an init_cfg ('Pretrained' checkpoints are absent offline)."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()


ModuleList = nn.ModuleList


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg) if cfg is not None else dict(type="Conv2d")
    t = cfg.pop("type")
    cls = CONV_LAYERS.get(t)
    if cls is None:
        raise KeyError(f"conv layer {t} is not registered")
    if kwargs.get("bias") == "auto":
        kwargs["bias"] = True
    return cls(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cls = NORM_LAYERS.get(t)
    if cls is None:
        raise KeyError(f"norm layer {t} is not registered")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    abbr = "ln" if isinstance(layer, nn.LayerNorm) else "bn"
    return abbr + str(postfix), layer


def build_activation_layer(cfg):
    cfg = dict(cfg)
    cls = ACTIVATION_LAYERS.get(cfg.pop("type"))
    return cls(**cfg)


class ConvModule(nn.Module):
    """conv -> norm -> act (mmcv order default); bias='auto' means bias = (norm_cfg is None)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                     dilation=dilation, groups=groups, bias=bias)
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act = dict(act_cfg)
            if act["type"] == "ReLU":
                act.setdefault("inplace", inplace)
            self.activate = build_activation_layer(act)

    def forward(self, x):
        if self.with_norm and isinstance(self.conv, nn.Conv2d):
            from ..dense import _foldable, conv_bn_act, fusable
            norm = getattr(self, self.norm_name)
            relu = self.with_activation and isinstance(self.activate, nn.ReLU)
            if _foldable(norm) and fusable(x) and (relu or not self.with_activation):
                return conv_bn_act(self.conv, norm, relu, x)
        if isinstance(self.conv, nn.Conv2d):
            from .. import train_conv
            x = train_conv.conv2d(self.conv, x)   # training: the 3x3 layers on srf_wino43 (forward and data gradient)
        else:
            x = self.conv(x)
        if self.with_norm:
            from .. import train_conv
            norm = getattr(self, self.norm_name)
            x = norm(train_conv.bn_train_input(norm, x))   # a train-mode BatchNorm never sees a channels-last tensor (MIOpen crash)
        if self.with_activation:
            x = self.activate(x)
        return x


def force_fp32(*a, **k):
    """mmcv.runner.force_fp32: everything on this path already runs in fp32."""
    def deco(f):
        return f
    return deco


auto_fp16 = force_fp32
