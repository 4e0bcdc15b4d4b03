"""`SECONDCustom` (mmdet3d_plugin/models/backbones/second_custom.py:10-91): dense conv3x3-BN-ReLU stacks on the
BEV map.  Plain torch modules own the parameters (module names `blocks.{i}.{j}` as in the reference); fp32 GPU inference
runs them channels-last on the Winograd kernel of csrc/conv.hip (srfdet3d_amd/nhwc.py), everything else on MIOpen."""
from torch import nn

from ..compat.cnn import BaseModule, build_conv_layer, build_norm_layer
from ..compat.registry import BACKBONES
from ..dense import run_sequential


@BACKBONES.register_module()
class SECONDCustom(BaseModule):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False),
                 init_cfg=None, pretrained=None):
        super().__init__(init_cfg=init_cfg)
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        widths_in = [in_channels, *out_channels[:-1]]
        stages = []
        for cin, cout, n, stride in zip(widths_in, out_channels, layer_nums, layer_strides):
            layers = []
            for j in range(n + 1):
                layers += [build_conv_layer(conv_cfg, cin if j == 0 else cout, cout, 3,
                                            stride=stride if j == 0 else 1, padding=1),
                           build_norm_layer(norm_cfg, cout)[1], nn.ReLU(inplace=True)]
            stages.append(nn.Sequential(*layers))
        self.blocks = nn.ModuleList(stages)

    def forward(self, x):
        from .. import nhwc
        if nhwc.enabled() and nhwc.second_supported(self, x):
            return nhwc.second_forward(self, x)   # fp32 inference on the GPU: channels-last, 3x3 / stride-1 layers on srf_wino43 (F(4x4, 3x3)), stride-2 on srf_conv_gemm_nhwc
        outs = []
        for stage in self.blocks:
            x = run_sequential(stage, x)
            outs.append(x)
        return tuple(outs)
