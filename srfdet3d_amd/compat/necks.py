"""mmdet 2.x FPN as the reference configs use it (configs/nus/srfdet_voxel_nusc_L.py:67-76;
SURVEY.md Appendix B.6): laterals 1x1, top-down nearest upsample-add, 3x3 outs, extra stride-2 convs."""
import torch.nn.functional as F
from torch import nn

from .. import ops
from .cnn import BaseModule, ConvModule
from .registry import NECKS


@NECKS.register_module()
class FPN(BaseModule):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__(init_cfg)
        self.in_channels = list(in_channels)
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.upsample_cfg = dict(upsample_cfg)
        self.backbone_end_level = self.num_ins if end_level in (-1, self.num_ins - 1) else end_level + 1
        self.start_level = start_level
        if add_extra_convs is True:
            add_extra_convs = "on_input"
        self.add_extra_convs = add_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg, act_cfg=act_cfg,
                                                 inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))
        extra = num_outs - self.backbone_end_level + start_level
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = in_channels[self.backbone_end_level - 1] if (i == 0 and add_extra_convs == "on_input") \
                    else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))

    def forward(self, inputs):
        from .. import nhwc
        if nhwc.enabled() and nhwc.fpn_supported(self, inputs):
            return nhwc.fpn_forward(self, inputs)   # channels-last levels from the NHWC backbone path
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        n = len(lat)
        for i in range(n - 1, 0, -1):
            if "scale_factor" in self.upsample_cfg:
                lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], **self.upsample_cfg)
            elif self.upsample_cfg.get("mode", "nearest") == "nearest" and len(self.upsample_cfg) == 1 and \
                    ops.upsample_add_supported(lat[i - 1], lat[i]):
                lat[i - 1] = ops.upsample_add(lat[i - 1], lat[i])  # one pass instead of upsample-to-temporary + add
            else:
                lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], **self.upsample_cfg)
        outs = [self.fpn_convs[i](lat[i]) for i in range(n)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - n):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                src = {"on_input": inputs[self.backbone_end_level - 1], "on_lateral": lat[-1],
                       "on_output": outs[-1]}[self.add_extra_convs]
                outs.append(self.fpn_convs[n](src))
                for i in range(n + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)
