"""`RoIAlign`, `SingleRoIExtractor` and `bbox2roi` with the mmcv / mmdet call surface, on the HIP gather kernel.

Reference call sites: `build_roi_extractor(roi_extractor_lidar)` at mmdet3d_plugin/models/sparse_heads/
srfdet_head.py:143,175 (config configs/nus/srfdet_voxel_nusc_LC.py:169-178) and `pooler(feats[:num_inputs], rois)`
at :1685, :2548, :2626; `bbox2roi` at :1683, :2521-2526.
"""
import torch
from torch import nn

from . import ops
from .compat.registry import ROI_EXTRACTORS, ROI_LAYERS


def bbox2roi(bbox_list):
    """list of (n_i, 4+) boxes per sample -> (sum n_i, 5) [batch_idx, x1, y1, x2, y2]."""
    rois = []
    for i, b in enumerate(bbox_list):
        if b.size(0) > 0:
            rois.append(torch.cat([b.new_full((b.size(0), 1), i), b[:, :4]], dim=-1))
        else:
            rois.append(b.new_zeros((0, 5)))
    return torch.cat(rois, 0)


@ROI_LAYERS.register_module()
class RoIAlign(nn.Module):
    """mmcv.ops.RoIAlign(avg pooling).  `aligned=True` is the mmcv default and what mmdet builds."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True, use_torchvision=False):
        super().__init__()
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        if not isinstance(output_size, int):
            assert output_size[0] == output_size[1]
        assert pool_mode == "avg" and aligned and sampling_ratio > 0, "only the mode the reference configs use"
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)

    def forward(self, feat, rois):
        # a single level is the multi-level gather with one entry
        return ops.roi_extract([feat], rois, [1.0 / self.spatial_scale], self.output_size, self.sampling_ratio,
                               finest_scale=1e30)


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    """mmdet SingleRoIExtractor: each RoI is pooled from the one level its scale maps to.
    The level assignment and the per-level pooling run in one launch."""

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        assert cfg.pop("type") == "RoIAlign"
        self.roi_layers = nn.ModuleList([RoIAlign(spatial_scale=1.0 / s, **cfg) for s in featmap_strides])
        self.out_channels = out_channels
        self.featmap_strides = list(featmap_strides)
        self.finest_scale = finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def forward(self, feats, rois, roi_scale_factor=None, out=None, accumulate=False, bin_major=False, n_sum=1):
        assert roi_scale_factor is None
        layer = self.roi_layers[0]
        return ops.roi_extract(list(feats[:self.num_inputs]), rois, self.featmap_strides, layer.output_size,
                               layer.sampling_ratio, float(self.finest_scale), out=out, accumulate=accumulate,
                               bin_major=bin_major, n_sum=n_sum)
