"""`Voxelization` and `DynamicScatter` with the mmcv.ops call surface, on the HIP kernels.

Reference call sites: `Voxelization(**pts_voxel_layer)` at mmdet3d_plugin/models/detectors/srfdet.py:58, used at
:218-247; `DynamicScatter(voxel_size, point_cloud_range, average_points)` at
mmdet3d_plugin/models/voxel_encoders/voxel_encoder.py:82, :99-102.
"""
import torch
from torch import nn

from . import ops


class Voxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.max_num_points = int(max_num_points)
        self.max_voxels = tuple(max_voxels) if isinstance(max_voxels, (tuple, list)) else (max_voxels, max_voxels)
        self.deterministic = deterministic  # results are always deterministic here
        gx, gy, gz = ops.grid_size(self.voxel_size, self.point_cloud_range)
        self.grid_size = [gx, gy, gz]
        self.pcd_shape = [gx, gy, gz + 1][::-1]
        # set by the caller that wants the per-voxel mean (HardSimpleVFE) out of the same kernel
        self.fused_mean_features = 0

    def forward(self, points):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        if self.max_num_points == -1:
            return ops.dynamic_voxelize(points, self.voxel_size, self.point_cloud_range)
        if max_voxels == -1:
            max_voxels = points.shape[0]
        voxels, coors, num, mean = ops.hard_voxelize(points, self.voxel_size, self.point_cloud_range,
                                                     self.max_num_points, max_voxels, self.fused_mean_features)
        if mean is not None:
            voxels.srf_vfe_mean = mean
        return voxels, coors, num

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range={self.point_cloud_range}, "
                f"max_num_points={self.max_num_points}, max_voxels={self.max_voxels})")


class DynamicScatter(nn.Module):
    """(feats (N,C), coors (N,4) b,z,y,x) -> (voxel_feats (M,C), voxel_coors (M,4)) in sorted voxel order.

    The sorted-unique pass depends only on `coors`, so its result (a `VoxelMap`) is cached on the coors tensor and
    shared by every DynamicScatter that is handed the same coordinates (the reference recomputes it 2-3 times per
    frame).  `last_map` exposes the point->voxel indices for gather-back."""

    def __init__(self, voxel_size, point_cloud_range, average_points):
        super().__init__()
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.average_points = bool(average_points)
        gx, gy, gz = ops.grid_size(self.voxel_size, self.point_cloud_range)
        self.grid_zyx = [gz, gy, gx]
        self.last_map = None
        self.static_rows = None  # set by a caller that needs fixed shapes (graphs.GraphedFrame): see ops.VoxelMap

    def voxel_map(self, coors, batch_size=None):
        cached = getattr(coors, "srf_voxel_map", None)
        if cached is not None and cached[0] == coors._version:
            return cached[1]
        c = coors if coors.dtype == torch.int32 else coors.int()
        if batch_size is None:
            batch_size = 1 if self.static_rows is not None else (int(c[:, 0].max().item()) + 1 if c.shape[0] else 1)
        vm = ops.VoxelMap(c, self.grid_zyx, max(batch_size, 1), static_rows=self.static_rows)
        if self.static_rows is not None:
            self.last_map_static = vm
        coors.srf_voxel_map = (coors._version, vm)
        return vm

    def forward(self, points, coors, batch_size=None):
        vm = self.voxel_map(coors, batch_size)
        self.last_map = vm
        return vm.reduce(points, "mean" if self.average_points else "max"), vm.coors
