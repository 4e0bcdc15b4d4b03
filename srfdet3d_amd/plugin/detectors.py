"""`SRFDet` / `SRFDetWaymo` detectors (mmdet3d_plugin/models/detectors/srfdet.py:13-343, srfdetwaymo.py:6-41):
voxelize -> voxel encoder -> sparse middle encoder -> SECOND -> FPN -> SRFDetHead.  Same registry names,
constructor arguments, sub-module attribute names and forward signatures as the reference."""
import torch
import torch.nn.functional as F

from .. import nhwc
from ..compat.boxes import bbox3d2result
from ..compat.cnn import BaseModule
from ..compat.registry import (DETECTORS, build_backbone, build_head, build_middle_encoder, build_neck,
                               build_voxel_encoder)
from ..voxel_layer import Voxelization
from .voxel_encoders import HardSimpleVFE


@DETECTORS.register_module()
class SRFDet(BaseModule):
    def __init__(self, use_img=False, freeze_img=True, use_grid_mask=False, img_backbone=None, img_neck=None,
                 pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None, pts_backbone=None,
                 pts_neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.use_img = use_img
        self.freeze_img = freeze_img
        self.use_grid_mask = use_grid_mask
        if use_img:
            self.img_backbone = build_backbone(img_backbone) if img_backbone is not None else None
            self.img_neck = build_neck(img_neck) if img_neck is not None else None
            self.grid_mask = None  # train-time augmentation only; identity in eval (grid_mask.py:89)
        if pts_voxel_layer:
            self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
            self.pts_voxel_layer_cfg = pts_voxel_layer
        if pts_voxel_encoder:
            self.pts_voxel_encoder = build_voxel_encoder(pts_voxel_encoder)
            if isinstance(self.pts_voxel_encoder, HardSimpleVFE) and pts_voxel_layer:
                # the voxelization kernel can emit the per-voxel mean in the same pass
                self.pts_voxel_layer.fused_mean_features = self.pts_voxel_encoder.num_features
        if pts_middle_encoder:
            self.pts_middle_encoder = build_middle_encoder(pts_middle_encoder)
        if pts_backbone:
            self.pts_backbone = build_backbone(pts_backbone)
        if pts_neck is not None:
            self.pts_neck = build_neck(pts_neck)
        else:
            self.pts_neck = None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg, use_img=use_img)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self._graphed_tail = None
        self._graphed_img = None
        self._graphed_frame = None
        # None = fp32 (the configs' default).  torch.float16 / torch.bfloat16 run the image backbone + neck under
        # autocast with fp32 outputs, i.e. the reference's `auto_fp16(apply_to=('img'), out_fp32=True)` mode
        # (srfdet.py:141); opt-in, never the default.
        self.img_autocast_dtype = None

    def enable_hip_graphs(self, enabled=True, img_overlap=False, whole_frame=True):
        """Replay the static-shape tail (SECOND -> FPN -> decoder -> decode) as a captured hipGraph in `simple_test`
        (see srfdet3d_amd/graphs.py), and with images the VoVNet -> FPN branch as a second graph.  Results are identical
        to the eager path; opt-in because a graph pins its buffers for the lifetime of the model.  img_overlap=True
        replays the image graph on a side stream beside the eager LiDAR half."""
        from ..graphs import GraphedFrame, GraphedImageBranch, GraphedTail
        self._graph_cfg = dict(enabled=enabled, img_overlap=img_overlap, whole_frame=whole_frame)
        self._graphed_tail = GraphedTail(self) if enabled else None
        self._graphed_img = GraphedImageBranch(self, overlap=img_overlap) if (enabled and self.use_img) else None
        # hard voxelization: the whole LiDAR frame replays as one graph (no host read-back inside the frame); with
        # cameras it reads the image graph's feature buffers in place
        self._graphed_frame = GraphedFrame(self) if (enabled and whole_frame and GraphedFrame.eligible(self)) else None
        self._graph_weights_sig = self._weights_signature()
        return self

    def _drop_derived_state(self):
        """Packed-weight images cached on the convolutions AND the captured hipGraphs.  A graph holds the raw device pointers of
        the packed weights of its warm-up pass; once those tensors are dropped their blocks return to the caching allocator, and a
        replay would read whatever lives there next as weights.  So both go together: the next inference pass packs
        afresh and recaptures."""
        from .. import nhwc
        nhwc.invalidate_caches(self)
        cfg = getattr(self, "_graph_cfg", None)
        if cfg is not None and (self._graphed_tail is not None or self._graphed_img is not None or self._graphed_frame is not None):
            self._graphed_tail = self._graphed_img = self._graphed_frame = None   # release the old graphs and their pools first
            self.enable_hip_graphs(**cfg)

    def train(self, mode=True):
        """Also drops what was derived from the weights (`_drop_derived_state`): whatever was done to them before a mode switch --
        incl. in-place updates through `.data`, which the caches' (version, pointer) key cannot see -- the next inference pass
        packs them afresh and recaptures its graphs.  A call that changes nothing (`.eval()` on a model already in eval mode whose
        parameters still are the tensors the graphs were captured with, at the same versions) keeps them."""
        if bool(mode) != self.training or self._weights_signature() != getattr(self, "_graph_weights_sig", None):
            self._drop_derived_state()
        return super().train(mode)

    def _weights_signature(self):
        """(sum of versions, sum of pointers, count) of the parameters and buffers: changes with every optimiser step, `copy_`,
        `load_state_dict`, `.to()` / `.float()` (new storage); NOT with updates through `.data` -- call `weights_changed()` after those."""
        v = p = n = 0
        for t in list(self.parameters()) + list(self.buffers()):
            v += t._version
            p += t.data_ptr()
            n += 1
        return (v, p, n)

    def weights_changed(self):
        """Tell the model that its weights were modified in place while it stayed in eval mode (optimizer.step() between two
        inference calls, `p.data.copy_()`, EMA updates): captured hipGraphs replay raw pointers to packed-weight images that no
        longer match.  `train()`, `load_state_dict()` and `.to()` / `.cuda()` / `.float()` call this themselves."""
        self._drop_derived_state()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_graph_cfg", None) is not None:
            self._drop_derived_state()   # the parameters may live in new storage: every captured pointer is stale
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._drop_derived_state()
        return out

    def init_weights(self):
        super().init_weights()
        if self.freeze_img and self.use_img and self.img_backbone is not None:
            for p in self.img_backbone.parameters():
                p.requires_grad = False

    # ---- entry points (srfdet.py:91-139, :278-340) -----------------------------------------------------------
    def forward(self, img=None, points=None, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, points, **kwargs)
        return self.forward_test(img, points, **kwargs)

    def forward_train(self, img, points, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_3d_ignore=None):
        img_feats, point_feats = self.extract_feat(img, points, img_metas)
        return self.bbox_head.forward_train(img_feats, point_feats, gt_bboxes_3d, gt_labels_3d, gt_bboxes_3d_ignore,
                                            img_metas)

    def forward_test(self, img, points, img_metas, **kwargs):
        if img is None:
            img = [img]
        elif points is None:
            points = [points]
        return self.simple_test(img[0], points[0], img_metas[0], **kwargs)

    def _test_bboxes(self, img, points, img_metas):
        from ..graphs import GraphValidationError
        try:
            return self._test_bboxes_graphs(img, points, img_metas)
        except GraphValidationError as err:
            import warnings
            warnings.warn(f"srfdet3d_amd: {err}")
            self._graphed_tail = self._graphed_img = self._graphed_frame = None
            return self._test_bboxes_graphs(img, points, img_metas)

    def _test_bboxes_graphs(self, img, points, img_metas):
        if (self._graphed_frame is not None and not self.training and points is not None and len(points) == 1
                and (img is None or self._graphed_img is not None)):
            img_feats = img_done = None
            if img is not None:
                img_feats, img_done = self._graphed_img(img, img_metas)
            return self._finish(*self._graphed_frame(points[0], img_metas, img_feats, img_done), img_metas)
        if self._graphed_tail is not None and not self.training and points is not None:
            img_static = False
            if img is not None and self._graphed_img is not None:
                # image branch: hipGraph on a side stream, overlapping the eager LiDAR half below
                img_feats, img_done = self._graphed_img(img, img_metas)
                img_static = True
            else:
                img_feats = self.extract_img_feat(img, img_metas) if img is not None else None
            bev = self.extract_bev(points)
            if img_static:
                torch.cuda.current_stream().wait_event(img_done)
            return self._finish(*self._graphed_tail(bev, img_feats, img_metas, img_static=img_static), img_metas)
        img_feats, point_feats = self.extract_feat(img, points, img_metas)
        return self.bbox_head.simple_test_bboxes(img_feats, point_feats, img_metas)

    def _finish(self, scores, boxes, sel, img_metas):
        """Detections from a graph replay: the NMS ran inside the graph with fixed shapes (`select_static`); one copy brings
        the survivors to the host.  Falls back to the eager selection if a sample overflowed the static capacity."""
        if sel is not None:
            res = self.bbox_head.results_from_static(sel[0].cpu(), sel[1].cpu(), img_metas)
            if res is not None:
                return res
        return self.bbox_head.get_bboxes(None, None, img_metas, decoded=(scores, boxes))

    def simple_test(self, img, points, img_metas, rescale=False):
        bbox_list = self._test_bboxes(img, points, img_metas)
        return [dict(pts_bbox=bbox3d2result(b, s, l)) for b, s, l in bbox_list]

    # ---- features ------------------------------------------------------------------------------------------------
    def extract_feat(self, img, points, img_metas=None):
        img_feats = self.extract_img_feat(img, img_metas) if img is not None else None
        point_feats = self.extract_point_features(points) if points is not None else None
        return img_feats, point_feats

    def extract_img_feat(self, img, img_metas):
        """(B, N, 3, H, W) -> list of (B, N, C, h, w) per pyramid level (srfdet.py:175-202)."""
        B = img.size(0)
        for meta in img_metas:
            meta.update(input_shape=img.shape[-2:])
        if img.dim() == 5:
            img = img.reshape(-1, *img.shape[2:])
        with torch.autocast(img.device.type, dtype=self.img_autocast_dtype, enabled=self.img_autocast_dtype is not None):
            feats = self.img_backbone(img)
            if isinstance(feats, dict):
                feats = list(feats.values())
            if self.img_neck is not None:
                feats = self.img_neck(feats)
        out = [f.float().view(B, f.shape[0] // B, *f.shape[1:]) for f in feats]
        return nhwc.ConsumedLevels(out) if isinstance(feats, nhwc.ConsumedLevels) else out

    @torch.no_grad()
    def voxelize(self, points):
        """list of (N_i, C) point clouds -> hard: (voxels, num_points, coors (M,4) b,z,y,x); dynamic: (points, coors (sum N,4))
        (srfdet.py:204-247)."""
        if self.pts_voxel_layer_cfg["max_num_points"] != -1:
            voxels, coors, nums = [], [], []
            for i, res in enumerate(points):
                v, c, n = self.pts_voxel_layer(res)
                voxels.append(v)
                nums.append(n)
                coors.append(F.pad(c, (1, 0), mode="constant", value=i))
            if len(points) == 1:
                return voxels[0], nums[0], coors[0]
            allv = torch.cat(voxels, 0)
            means = [getattr(v, "srf_vfe_mean", None) for v in voxels]
            if all(m is not None for m in means):  # keep the per-voxel means the voxelization kernel produced
                allv.srf_vfe_mean = torch.cat(means, 0)
            return allv, torch.cat(nums, 0), torch.cat(coors, 0)
        coors = [F.pad(self.pts_voxel_layer(res), (1, 0), mode="constant", value=i) for i, res in enumerate(points)]
        if len(points) == 1:
            return points[0], coors[0]
        return torch.cat(points, 0), torch.cat(coors, 0)

    def extract_bev_static(self, points, static_caps):
        """One sample, fixed shapes (see graphs.GraphedFrame): `points` (N_cap, C) with out-of-range filler rows ->
        (bev, [(name, device count, capacity), ...]); nothing is read back to the host."""
        from .. import ops
        vl = self.pts_voxel_layer
        if vl.max_num_points != -1:  # hard voxelization, mean fused in
            max_voxels = vl.max_voxels[0] if self.training else vl.max_voxels[1]
            if max_voxels == -1:
                max_voxels = points.shape[0]
            voxels, coors, num, mean, vnum = ops.hard_voxelize(points, vl.voxel_size, vl.point_cloud_range, vl.max_num_points,
                                                               max_voxels, vl.fused_mean_features, static=True)
            if mean is not None:                    # coors: (rows, 4) (0, z, y, x); padding rows -1
                voxels.srf_vfe_mean = mean
            voxel_features = self.pts_voxel_encoder(voxels, num, coors)
            rows = voxels.shape[0]
        else:  # dynamic voxelization + DynamicVFECustom: the scatter maps run at a fixed voxel capacity
            pc = ops.dynamic_voxelize(points, vl.voxel_size, vl.point_cloud_range)
            batch = torch.where(pc[:, :1] < 0, -1, 0).to(pc.dtype)
            pt_coors = torch.cat([batch, pc], dim=1)
            enc = self.pts_voxel_encoder
            scatters = [enc.scatter, enc.vfe_scatter, enc.cluster_scatter]
            rows = int(static_caps["__voxels__"])
            for sc in scatters:
                sc.static_rows = rows
            try:
                voxel_features, coors = enc(points, pt_coors)
                vnum = enc.cluster_scatter.last_map_static.num_dev
            finally:
                for sc in scatters:
                    sc.static_rows = None
        caps = {k: v for k, v in static_caps.items() if k != "__voxels__"}
        bev, counts = self.pts_middle_encoder(voxel_features, coors, 1, static_caps=dict(caps, __rows__=vnum))
        return bev, [("voxels", vnum, rows)] + counts

    def extract_point_features(self, points):
        x = self.pts_backbone(self.extract_bev(points))
        if self.pts_neck is not None:
            x = self.pts_neck(x)
        return x

    def extract_bev(self, points):
        """points -> dense BEV map (B, C*D, H, W): the data-dependent part of the path (voxelize, VFE, sparse encoder)."""
        batch_size = len(points)
        if self.pts_voxel_layer_cfg["max_num_points"] != -1:
            voxels, num_points, coors = self.voxelize(points)
            voxel_features = self.pts_voxel_encoder(voxels, num_points, coors)
        else:
            pts, pt_coors = self.voxelize(points)
            voxel_features, coors = self.pts_voxel_encoder(pts, pt_coors)
        # the reference derives batch_size from coors[-1, 0] + 1 (srfdet.py:258, :271), a device->host sync; the
        # number of point clouds handed in is the same number whenever the last sample has at least one voxel
        return self.pts_middle_encoder(voxel_features, coors, batch_size)


@DETECTORS.register_module()
class SRFDetWaymo(SRFDet):
    """Same model; results are returned without the `pts_bbox` wrapper (srfdetwaymo.py:13-41)."""

    def _finish(self, scores, boxes, sel, img_metas):
        """Detections from a graph replay: the NMS ran inside the graph with fixed shapes (`select_static`); one copy brings
        the survivors to the host.  Falls back to the eager selection if a sample overflowed the static capacity."""
        if sel is not None:
            res = self.bbox_head.results_from_static(sel[0].cpu(), sel[1].cpu(), img_metas)
            if res is not None:
                return res
        return self.bbox_head.get_bboxes(None, None, img_metas, decoded=(scores, boxes))

    def simple_test(self, img, points, img_metas, rescale=False):
        return [bbox3d2result(b, s, l) for b, s, l in self._test_bboxes(img, points, img_metas)]
