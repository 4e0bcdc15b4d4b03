"""Voxel feature encoders on the SRFDet3D path.

`HardSimpleVFE` is mmdet3d's (configs/nus/srfdet_voxel_nusc_L.py:40).
`DynamicVFECustom` / `DynamicVFELayer` mirror mmdet3d_plugin/models/voxel_encoders/voxel_encoder.py:10-240 and
utils.py:8-45 (same constructor arguments, same parameter names).  Differences that do not change results:
the sorted-unique voxel map is computed ONCE per call and shared by the cluster / VFE scatters, and
"map voxel value back to its points" is an index with that map instead of a dense Z*Y*X*B int64 canvas
(voxel_encoder.py:118-158 allocates 0.7 GB on the KITTI grid).
"""
import torch
from torch import nn
from torch.nn import functional as F

from ..compat.cnn import build_norm_layer
from ..compat.registry import VOXEL_ENCODERS
from ..voxel_layer import DynamicScatter


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors=None):
        """(M, max_points, C) zero-padded points, (M,) counts -> (M, num_features) mean."""
        fused = getattr(features, "srf_vfe_mean", None)
        if fused is not None and fused.shape[1] == self.num_features:
            return fused  # produced by the voxelization kernel in the same pass
        s = features[:, :, :self.num_features].sum(dim=1, keepdim=False)
        return (s / num_points.type_as(features).view(-1, 1)).contiguous()


class DynamicVFELayer(nn.Module):
    """Linear(no bias) -> norm -> ReLU on per-point features (utils.py:8-45)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        return F.relu(self.norm(self.linear(inputs)))


@VOXEL_ENCODERS.register_module()
class DynamicVFECustom(nn.Module):
    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", fusion_layer=None,
                 return_point_feats=False, with_centroid_aware_vox=True, centroid_to_point_pos_emb_dims=32):
        super().__init__()
        assert mode in ("avg", "max") and len(feat_channels) > 0
        assert fusion_layer is None, "point-wise image fusion is not used by any SRFDet3D config"
        if with_centroid_aware_vox:
            in_channels += centroid_to_point_pos_emb_dims
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self._with_centroid_aware_vox = with_centroid_aware_vox
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.vx, self.vy, self.vz = voxel_size
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range
        self.scatter = DynamicScatter(voxel_size, point_cloud_range, True)
        widths = [self.in_channels] + list(feat_channels)
        self.vfe_layers = nn.ModuleList([
            DynamicVFELayer(widths[i] * (2 if i > 0 else 1), widths[i + 1], norm_cfg) for i in range(len(widths) - 1)])
        self.num_vfe = len(self.vfe_layers)
        self.vfe_scatter = DynamicScatter(voxel_size, point_cloud_range, mode != "max")
        self.cluster_scatter = DynamicScatter(voxel_size, point_cloud_range, average_points=True)
        self.fusion_layer = None
        if with_centroid_aware_vox:
            d = centroid_to_point_pos_emb_dims
            self.cen2point_pos_enc = nn.Sequential(nn.Linear(3, d, bias=False), nn.BatchNorm1d(d), nn.Tanh(),
                                                   nn.Linear(d, d, bias=False), nn.BatchNorm1d(d), nn.Tanh())

    @staticmethod
    def _to_points(voxel_values, point2voxel):
        """value of each point's voxel; points outside the range (map -1) read row 0 like the reference's canvas."""
        return voxel_values[point2voxel.clamp(min=0).long()]

    def map_voxel_center_to_point(self, pts_coors, voxel_mean, voxel_coors):
        vm = self.cluster_scatter.voxel_map(pts_coors)
        return self._to_points(voxel_mean, vm.point2voxel)

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None):
        """(N,C) points, (N,4) int (b,z,y,x) with -1 rows for dropped points -> ((M,C') voxel feats, (M,4) coors)."""
        vm = self.cluster_scatter.voxel_map(coors)
        parts = [features]
        if self._with_cluster_center:
            voxel_mean = vm.reduce(features, "mean")
            f_cluster = features[:, :3] - self._to_points(voxel_mean, vm.point2voxel)[:, :3]
            if self._with_centroid_aware_vox:
                f_cluster = self.cen2point_pos_enc(f_cluster)
            parts.append(f_cluster)
        if self._with_voxel_center:
            cf = coors.type_as(features)
            parts.append(torch.stack([features[:, 0] - (cf[:, 3] * self.vx + self.x_offset),
                                      features[:, 1] - (cf[:, 2] * self.vy + self.y_offset),
                                      features[:, 2] - (cf[:, 1] * self.vz + self.z_offset)], dim=1))
        if self._with_distance:
            parts.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        x = torch.cat(parts, dim=-1)
        mode = "mean" if self.vfe_scatter.average_points else "max"
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(x)
            voxel_feats = vm.reduce(point_feats, mode)
            if i != self.num_vfe - 1:
                x = torch.cat([point_feats, self._to_points(voxel_feats, vm.point2voxel)], dim=1)
        if self.return_point_feats:
            return point_feats
        return voxel_feats, vm.coors
