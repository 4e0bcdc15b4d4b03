"""Pillar-config encoder: `PillarFeatureNetCustom` / `PFNLayer`
(mmdet3d_plugin/models/voxel_encoders/pillar_encoder_custom.py:13-160, utils.py:63-146) and mmdet3d's
`PointPillarsScatter` (configs/nus/srfdet_pillar_nusc_L.py:53-54).  Pillars come from the hard-voxelization kernel with
20 points per pillar on a 512 x 512 grid; the scatter to the BEV canvas is the densify kernel with D = 1."""
import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ..compat.cnn import BaseModule, build_norm_layer
from ..compat.registry import MIDDLE_ENCODERS, VOXEL_ENCODERS


class PFNLayer(nn.Module):
    """Linear(no bias) -> BN over channels -> ReLU -> max/avg over the points of a pillar (utils.py:63-146)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), last_layer=False,
                 mode="max"):
        super().__init__()
        assert mode in ("max", "avg")
        self.fp16_enabled = False
        self.name = "PFNLayer"
        self.last_vfe = last_layer
        self.units = out_channels if last_layer else out_channels // 2
        self.norm = build_norm_layer(norm_cfg, self.units)[1]
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        self.mode = mode

    def forward(self, inputs, num_voxels=None, aligned_distance=None):
        x = self.linear(inputs)                                        # (N, M, units)
        x = F.relu(self.norm(x.transpose(1, 2)).transpose(1, 2))       # BN1d over the channel axis
        if aligned_distance is not None:
            x = x * aligned_distance.unsqueeze(-1)
        if self.mode == "max":
            pooled = x.max(dim=1, keepdim=True)[0]
        else:
            pooled = x.sum(dim=1, keepdim=True) / num_voxels.type_as(inputs).view(-1, 1, 1)
        if self.last_vfe:
            return pooled
        return torch.cat([x, pooled.expand(-1, inputs.shape[1], -1)], dim=2)


@VOXEL_ENCODERS.register_module()
class PillarFeatureNetCustom(BaseModule):
    def __init__(self, in_channels=4, feat_channels=(64,), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", legacy=True, init_cfg=None):
        super().__init__(init_cfg=init_cfg)
        assert len(feat_channels) > 0
        self.legacy = legacy
        in_channels += 3 * bool(with_cluster_center) + 3 * bool(with_voxel_center) + bool(with_distance)
        self._with_distance, self._with_cluster_center, self._with_voxel_center = with_distance, with_cluster_center, with_voxel_center
        self.fp16_enabled = False
        self.in_channels = in_channels
        widths = [in_channels] + list(feat_channels)
        self.pfn_layers = nn.ModuleList([PFNLayer(widths[i], widths[i + 1], norm_cfg=norm_cfg, last_layer=i == len(widths) - 2,
                                                  mode=mode) for i in range(len(widths) - 1)])
        self.vx, self.vy, self.vz = voxel_size
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range

    def forward(self, features, num_points, coors):
        """(N, M, C) zero-padded pillar points, (N,) counts, (N, 4) (b,z,y,x) -> (N, C_out) pillar features."""
        parts = [features]
        if self._with_cluster_center:
            mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(features).view(-1, 1, 1)
            parts.append(features[:, :, :3] - mean)
        if self._with_voxel_center:
            cf = coors.to(features.dtype)
            centre = torch.stack([cf[:, 3] * self.vx + self.x_offset, cf[:, 2] * self.vy + self.y_offset,
                                  cf[:, 1] * self.vz + self.z_offset], dim=1).unsqueeze(1)
            f_center = features[:, :, :3] - centre
            if self.legacy:  # the legacy path overwrote the xyz of `features` in place before concatenating
                parts[0] = torch.cat([f_center, features[:, :, 3:]], dim=2)
            parts.append(f_center)
        if self._with_distance:
            parts.append(torch.norm(features[:, :, :3], 2, 2, keepdim=True))
        x = torch.cat(parts, dim=-1)
        mask = torch.arange(x.shape[1], device=x.device, dtype=torch.int).view(1, -1) < num_points.int().view(-1, 1)
        x = x * mask.unsqueeze(-1).type_as(x)
        for pfn in self.pfn_layers:
            x = pfn(x, num_points)
        return x.squeeze(1)


@MIDDLE_ENCODERS.register_module()
class PointPillarsScatter(nn.Module):
    """mmdet3d PointPillarsScatter: (N, C) pillar features + (N, 4) coords -> (B, C, ny, nx) canvas."""

    def __init__(self, in_channels, output_shape):
        super().__init__()
        self.output_shape = output_shape
        self.ny, self.nx = output_shape
        self.in_channels = in_channels
        self.fp16_enabled = False

    def forward(self, voxel_features, coors, batch_size=None):
        if batch_size is None:
            batch_size = int(coors[:, 0].max().item()) + 1
        c = coors.int().clone()
        c[:, 1] = 0  # pillars: a single z slab
        return ops.densify(voxel_features, c.contiguous(), int(batch_size), [1, self.ny, self.nx]).view(
            int(batch_size), self.in_channels, self.ny, self.nx)
