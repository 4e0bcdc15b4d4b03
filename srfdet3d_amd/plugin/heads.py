"""The sparse-region-fusion decoder: `SRFDetHead`, `SingleSRFDetHeadLiDAR`, `SingleSRFDetHead`, `DynamicConv`.

Mirrors mmdet3d_plugin/models/sparse_heads/srfdet_head.py (SRFDetHead :48-1344, SingleSRFDetHeadLiDAR :1347-1689,
SingleSRFDetHead :2103-2630, DynamicConv :2633-2693): same registry names, constructor arguments, parameter names
(so checkpoints load by key) and call signatures.  What differs is how a stage is executed:

  * proposal box -> corners -> BEV / per-camera RoIs is ONE kernel (srf_box_rois) instead of ~60 small torch ops,
    and it keeps the reference's in-place overwrite of the box centres with metres (srfdet_head.py:1646, :2587),
    which apply_deltas relies on;
  * the RoI gather writes (R, 49, C) directly -- the layout DynamicConv multiplies -- from channels-last feature
    maps, so the permutes of :1480-1481 / :2257-2263 and :2667 disappear and every tap is a 512-byte read;
  * lidar2img is uploaded once per frame, not once per stage (:2452-2456);
  * the camera sum (:2551-2561) keeps the reference's roi indexing b + cam*bs against maps flattened b*n_cam + cam
    (SURVEY.md finding 7); `corrected_cam_indexing=True` switches to the consistent indexing and is off by default.
"""
import copy
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import nhwc, ops
from ..compat.cnn import BaseModule, ConvModule, ModuleList, build_activation_layer, build_conv_layer
from ..compat.registry import HEADS, LOSSES, BBOX_ASSIGNERS, build_head, build_roi_extractor
from ..dense import _fold_bn2d, _foldable, fusable, linear_graph_safe
from ..roi import SingleRoIExtractor
from .bbox_util import denormalize_bbox

_DEFAULT_SCALE_CLAMP = math.log(100000.0 / 16)


_CONSTS = {}


def _const(values, like):
    """small constant tensor on `like`'s device, uploaded once (a host->device copy per call would also be illegal
    inside a hipGraph capture)."""
    key = (tuple(float(v) for v in values), like.device, like.dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=like.dtype, device=like.device)
        _CONSTS[key] = t
    return t


_STAGE_SIDE = {}


def _stage_side_stream(device):
    """one side stream per device for the forked half of a decoder stage (`_StageBase._fork_gather`)"""
    key = (device.type, device.index)
    st = _STAGE_SIDE.get(key)
    if st is None:
        st = _STAGE_SIDE[key] = torch.cuda.Stream(device=device)
    return st


def _channels_last(feats):
    return [f if f.is_contiguous(memory_format=torch.channels_last) else ops.to_channels_last(f) for f in feats]


class DynamicConv(nn.Module):
    """Per-proposal two-layer 1x1 'dynamic' conv on the 7x7 RoI feature, then a 49C -> C projection
    (srfdet_head.py:2633-2693)."""

    def __init__(self, feat_channels, dynamic_dim=64, dynamic_num=2, pooler_resolution=7):
        super().__init__()
        self.feat_channels = feat_channels
        self.dynamic_dim = dynamic_dim
        self.dynamic_num = dynamic_num
        self.num_params = feat_channels * dynamic_dim
        self.dynamic_layer = nn.Linear(feat_channels, dynamic_num * self.num_params)
        self.norm1 = nn.LayerNorm(dynamic_dim)
        self.norm2 = nn.LayerNorm(feat_channels)
        self.activation = nn.ReLU(inplace=True)
        self.out_layer = nn.Linear(feat_channels * pooler_resolution ** 2, feat_channels)
        self.norm3 = nn.LayerNorm(feat_channels)

    def forward(self, prop_feats, roi_feats):
        """prop_feats (1,R,C) or (R,C); roi_feats (S,R,C) as the reference passes it, or (R,S,C) when
        `roi_feats.srf_bin_major` is set by the fused gather.  -> (R,C)."""
        C, d = self.feat_channels, self.dynamic_dim
        feats = roi_feats if getattr(roi_feats, "srf_bin_major", False) else roi_feats.permute(1, 0, 2)
        params = self.dynamic_layer(prop_feats.reshape(-1, C))
        w1 = params[:, :self.num_params].view(-1, C, d)
        w2 = params[:, self.num_params:].view(-1, d, C)
        x = F.relu(self.norm1(torch.bmm(feats, w1)))
        x = F.relu(self.norm2(torch.bmm(x, w2)))
        x = self.out_layer(x.flatten(1))
        return F.relu(self.norm3(x))


class _StageBase(BaseModule):
    """One refinement stage (shared by the LiDAR-only and the fusion class; attribute names as in the reference)."""

    def __init__(self, num_classes=80, feat_channels=256, pooler_resolution=7, use_focal_loss=True, use_fed_loss=False,
                 dim_feedforward=2048, num_cls_convs=1, num_reg_convs=3, num_heads=8, dropout=0.0,
                 scale_clamp=_DEFAULT_SCALE_CLAMP, bbox_weights=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2),
                 act_cfg=dict(type="ReLU", inplace=True), dynamic_conv=dict(dynamic_dim=64, dynamic_num=2),
                 pc_range=None, voxel_size=None, init_cfg=None, is_kitti=None, use_fusion=False):
        super().__init__(init_cfg)
        C = feat_channels
        self.feat_channels = self.feat_channels_lidar = C
        self.pc_range_lidar = pc_range
        self.voxel_size_lidar = voxel_size
        self.is_kitti = bool(is_kitti)
        self.use_fusion = use_fusion
        # False reproduces the reference's image-RoI batch ids (b + cam*bs) against maps flattened b*n_cam + cam, which
        # only coincide for bs = 1 (SURVEY.md finding 7); True makes the ids consistent for bs > 1
        self.corrected_cam_indexing = False
        self.self_attn_lidar = nn.MultiheadAttention(C, num_heads, dropout=dropout)
        self.inst_interact_lidar = DynamicConv(C, dynamic_conv["dynamic_dim"], dynamic_conv["dynamic_num"],
                                               pooler_resolution)
        self.linear1_lidar = nn.Linear(C, dim_feedforward)
        self.dropout_lidar = nn.Dropout(dropout)
        self.linear2_lidar = nn.Linear(dim_feedforward, C)
        self.norm1_lidar, self.norm2_lidar, self.norm3_lidar = nn.LayerNorm(C), nn.LayerNorm(C), nn.LayerNorm(C)
        self.dropout1_lidar, self.dropout2_lidar, self.dropout3_lidar = (nn.Dropout(dropout) for _ in range(3))
        self.activation_lidar = build_activation_layer(act_cfg)

        def tower(n):
            mods = []
            for _ in range(n):
                mods += [nn.Linear(C, C, False), nn.LayerNorm(C), nn.ReLU(inplace=True)]
            return ModuleList(mods)

        self.cls_module_lidar = tower(num_cls_convs)
        self.reg_module_lidar = tower(num_reg_convs)
        self.use_focal_loss, self.use_fed_loss = use_focal_loss, use_fed_loss
        self.class_logits_lidar = nn.Linear(C, num_classes if (use_focal_loss or use_fed_loss) else num_classes + 1)
        self.bboxes_delta_lidar = nn.Linear(C, len(bbox_weights))
        self.scale_clamp = scale_clamp
        self.bbox_weights = list(bbox_weights)
        self.fuse_stage_tail = True  # FFN + towers + deltas in one launch (ops.stage_tail) when the shapes allow
        if use_fusion:
            self.output_fused_proj = nn.Linear(2 * C, C)
        if init_cfg is None:
            self.init_cfg = [dict(type="Normal", std=0.01, override=dict(name="class_logits_lidar")),
                             dict(type="Normal", std=0.001, override=dict(name="bboxes_delta_lidar"))]

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        nn.init.constant_(self.class_logits_lidar.bias, -math.log((1 - 0.01) / 0.01))
        nn.init.constant_(self.bboxes_delta_lidar.bias.data[2:], 0.0)

    # ---- geometry + gather ----------------------------------------------------------------------------------
    def _gather(self, feats, rois, pooler):
        """RoI features as (R, S, C).  The fused path asks the extractor for that layout directly."""
        if isinstance(pooler, SingleRoIExtractor):
            lv = feats[:pooler.num_inputs]
            if torch.is_grad_enabled() and any(f.requires_grad for f in lv):
                layer = pooler.roi_layers[0]
                return ops.roi_extract_autograd(lv, rois, pooler.featmap_strides, layer.output_size, layer.sampling_ratio,
                                                float(pooler.finest_scale), bin_major=True)
            out = pooler(_channels_last(lv), rois, bin_major=True)
        else:  # any object with the mmdet call surface (feats, rois) -> (R, C, 7, 7)
            out = pooler(feats[:pooler.num_inputs], rois)
            out = out.flatten(2).permute(0, 2, 1).contiguous()
        return out

    def _geometry(self, bboxes, want_bev, l2i):
        """RoIs of the proposals and the boxes with centres in metres.  Inference overwrites the caller's tensor in place
        like the reference (srfdet_head.py:1646); when autograd is recording, the kernel runs on a detached copy (RoIs
        carry no gradient) and the metres are formed by differentiable torch ops instead."""
        if torch.is_grad_enabled() and bboxes.requires_grad:
            rois_bev, rois_img = ops.box_rois(bboxes.detach().clone().contiguous(), self.pc_range_lidar, self.voxel_size_lidar,
                                              mutate_centres=False, want_bev=want_bev, lidar2img=l2i)
            r = self.pc_range_lidar
            ctr = bboxes[..., :3] * _const([r[3] - r[0], r[4] - r[1], r[5] - r[2]], bboxes) + _const(r[:3], bboxes)
            return torch.cat([ctr, bboxes[..., 3:]], dim=-1), rois_bev, rois_img
        if not bboxes.is_contiguous():
            raise RuntimeError("srfdet3d_amd: proposal boxes must be contiguous (their centres are rewritten in place)")
        rois_bev, rois_img = ops.box_rois(bboxes, self.pc_range_lidar, self.voxel_size_lidar, mutate_centres=True,
                                          want_bev=want_bev, lidar2img=l2i)
        return bboxes, rois_bev, rois_img

    def points_feats_sampling_bboxes_roi(self, points_feats, bboxes, pooler, img_metas=None):
        """BEV RoI features of each proposal, (bs*n_p, C, 7, 7); overwrites bboxes[..., :3] with metres in place
        (srfdet_head.py:1627-1689 / :2568-2630)."""
        rois, _ = ops.box_rois(bboxes, self.pc_range_lidar, self.voxel_size_lidar, mutate_centres=True)
        r = self._gather(points_feats, rois, pooler)
        return r.permute(0, 2, 1).reshape(r.shape[0], r.shape[2], 7, 7)

    @staticmethod
    def _lidar2img(img_metas, like):
        static = img_metas[0].get("srf_lidar2img_static") if isinstance(img_metas[0], dict) else None
        if static is not None:
            return static  # a hipGraph's persistent buffer (graphs._StaticMetas keeps it current)
        m = np.asarray([meta["lidar2img"] for meta in img_metas], dtype=np.float32)
        if m.ndim == 3:  # KITTI: one camera, (bs,4,4) (srfdet_head.py:2462-2464)
            m = m[:, None]
        m = np.ascontiguousarray(m)
        # the device copy is cached in the metas (one upload per frame, not one per stage: :2452-2456) next to the host
        # matrices it was made from: a metas dict that comes back with other matrices or another batch size is re-uploaded
        cached = img_metas[0].get("srf_lidar2img_dev") if isinstance(img_metas[0], dict) else None
        if cached is not None and cached[0].device == like.device and cached[1].shape == m.shape and np.array_equal(cached[1], m):
            return cached[0]
        t = torch.from_numpy(m).to(like.device)
        if isinstance(img_metas[0], dict):
            img_metas[0]["srf_lidar2img_dev"] = (t, m)
        return t

    def _img_rois_feats(self, img_feats, rois_img, pooler_img, bs, n_p, n_cam):
        """gather over all cameras and sum them per proposal -> (bs*n_p, S, C) (srfdet_head.py:2543-2562)."""
        flat = [f.reshape(f.shape[0] * f.shape[1], *f.shape[2:]) for f in img_feats]
        if self.corrected_cam_indexing and bs > 1:
            ids = rois_img[:, 0]
            rois_img = rois_img.clone()
            rois_img[:, 0] = torch.remainder(ids, bs) * n_cam + torch.div(ids, bs, rounding_mode="floor")
        r = self._gather(flat, rois_img, pooler_img)  # (n_cam*bs*n_p, S, C), cam-major rows
        return r.view(n_cam, bs * n_p, r.shape[1], r.shape[2]).sum(dim=0)

    def img_feats_sampling_bboxes_roi(self, img_feats, bboxes, pooler, img_metas):
        """image RoI features summed over cameras, (bs*n_p, C, 7, 7); leaves `bboxes` untouched (:2424-2566)."""
        bs, n_p = bboxes.shape[:2]
        l2i = self._lidar2img(img_metas, bboxes)
        _, rois_img = ops.box_rois(bboxes.clone(), self.pc_range_lidar, self.voxel_size_lidar, mutate_centres=False,
                                   want_bev=False, lidar2img=l2i)
        r = self._img_rois_feats(img_feats, rois_img, pooler, bs, n_p, l2i.shape[1])
        return r.permute(0, 2, 1).reshape(r.shape[0], r.shape[2], 7, 7)

    # ---- stage arithmetic -----------------------------------------------------------------------------------
    def _hip_eligible(self, roi_feats):
        dc = self.inst_interact_lidar
        return (roi_feats.is_cuda and not torch.is_grad_enabled() and not self.training
                and (dc.feat_channels, dc.dynamic_dim) in ((128, 32), (256, 64)) and roi_feats.shape[1] <= 64
                and isinstance(self.activation_lidar, nn.ReLU) and dc.dynamic_num == 2)

    def _tail_fusable(self):
        """srf_stage_tail covers C == 128, FFN width a multiple of 128 up to 512 and towers of at most 4 layers."""
        F = self.linear1_lidar.weight.shape[0]
        return (self.fuse_stage_tail and self.feat_channels_lidar == 128 and F % 128 == 0 and F <= 512
                and len(self.cls_module_lidar) <= 12 and len(self.reg_module_lidar) <= 12
                and self.class_logits_lidar.weight.shape[0] <= 32 and 8 <= self.bboxes_delta_lidar.weight.shape[0] <= 32)

    def _attend_hip(self, q0, bs):
        """The half of a stage that needs no RoI feature (srfdet_head.py:1486-1496, :2636-2646 and the head of DynamicConv, :2671):
        self-attention among the proposals + norm1, then the dynamic parameters.  -> (q1, params)."""
        mha = self.self_attn_lidar
        qkv = ops.linear(q0, mha.in_proj_weight, mha.in_proj_bias)
        att = ops.self_attention(qkv, mha.num_heads, batch=bs)  # among the proposals of each sample, one launch for the batch
        q1 = ops.linear(att, mha.out_proj.weight, mha.out_proj.bias, residual=q0, ln2=self.norm1_lidar)
        dc = self.inst_interact_lidar
        return q1, ops.linear(q1, dc.dynamic_layer.weight, dc.dynamic_layer.bias)

    def _fork_gather(self, prop_feats, bs, n_p, gather, default_on):
        """Inside a graph capture the geometry + RoI gather of a stage (`gather()` -> (roi_feats, boxes_m)) and its attention half can be
        two branches of the graph: neither reads what the other writes (the gather reads the boxes and the feature maps, the attention
        half the proposal features), the gather fills 200 workgroups for 15-45 us, the attention half is a chain of four launches of
        1-64 workgroups.  Measured (round 5, same box, alternating runs): the fusion stages of LC, whose gather + projection take 69 us
        beside 42 us of attention, 33.12 -> 32.98 ms per frame; the LiDAR-only stages (23 us of gather) 3.66-3.67 -> 3.69 ms -- a fork /
        join pair costs about what the short branch saves.  So: on for the fusion stages, off for the LiDAR-only ones
        (SRF_STAGE_FORK=0 / 1 forces it).  -> (roi_feats, boxes_m, (q1, params) or None)."""
        C = self.feat_channels_lidar
        want = os.environ.get("SRF_STAGE_FORK")
        on = default_on if want is None else want != "0"
        if (prop_feats is None or not prop_feats.is_cuda or torch.is_grad_enabled() or self.training
                or not torch.cuda.is_current_stream_capturing() or not on):
            roi_feats, boxes_m = gather()
            return roi_feats, boxes_m, None
        main = torch.cuda.current_stream()
        side = _stage_side_stream(prop_feats.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            roi_feats, boxes_m = gather()
        attended = self._attend_hip(prop_feats.reshape(bs * n_p, C).contiguous(), bs) if self._hip_eligible(
            prop_feats.new_empty((1, 49, 1))) else None
        main.wait_stream(side)   # the join; everything the branch allocated stays referenced by the caller until the stage ends
        return roi_feats, boxes_m, attended

    def _refine_hip(self, roi_feats, boxes_m, prop_feats, bs, n_p, attended=None):
        """The stage on the hand-written kernels of csrc/decoder.hip (inference): 10 launches (12 with geometry + gather) instead of ~60."""
        C = self.feat_channels_lidar
        R = bs * n_p
        S = roi_feats.shape[1]
        if attended is None:
            q0 = (roi_feats.mean(dim=1) if prop_feats is None else prop_feats).reshape(R, C).contiguous()
            attended = self._attend_hip(q0, bs)
        q1, params = attended
        dc = self.inst_interact_lidar
        mid = ops.dynconv_mid(roi_feats, params, dc.norm1, dc.norm2)
        obj = ops.linear(mid.view(R, S * C), dc.out_layer.weight, dc.out_layer.bias, ln1=dc.norm3, relu1=True, residual=q1,
                         ln2=self.norm2_lidar)
        if self._tail_fusable():
            cls_layers = [(self.cls_module_lidar[i], self.cls_module_lidar[i + 1]) for i in range(0, len(self.cls_module_lidar), 3)]
            reg_layers = [(self.reg_module_lidar[i], self.reg_module_lidar[i + 1]) for i in range(0, len(self.reg_module_lidar), 3)]
            obj, logits, pred = ops.stage_tail(obj, (self.linear1_lidar, self.linear2_lidar), self.norm3_lidar, cls_layers,
                                               reg_layers, self.class_logits_lidar, self.bboxes_delta_lidar,
                                               boxes_m.reshape(R, -1), self.bbox_weights[:6], self.pc_range_lidar,
                                               self.scale_clamp)
            return logits.view(bs, n_p, -1), pred.view(bs, n_p, -1), obj.view(1, R, C)
        hid = ops.linear(obj, self.linear1_lidar.weight, self.linear1_lidar.bias, relu1=True)
        obj = ops.linear(hid, self.linear2_lidar.weight, self.linear2_lidar.bias, residual=obj, ln2=self.norm3_lidar)
        cls_f, reg_f = obj, obj
        for i in range(0, len(self.cls_module_lidar), 3):
            cls_f = ops.linear(cls_f, self.cls_module_lidar[i].weight, None, ln1=self.cls_module_lidar[i + 1], relu1=True)
        for i in range(0, len(self.reg_module_lidar), 3):
            reg_f = ops.linear(reg_f, self.reg_module_lidar[i].weight, None, ln1=self.reg_module_lidar[i + 1], relu1=True)
        logits = ops.linear(cls_f, self.class_logits_lidar.weight, self.class_logits_lidar.bias)
        deltas = ops.linear(reg_f, self.bboxes_delta_lidar.weight, self.bboxes_delta_lidar.bias)
        pred = ops.apply_deltas(deltas, boxes_m.reshape(R, -1), self.bbox_weights[:6], self.pc_range_lidar, self.scale_clamp)
        return logits.view(bs, n_p, -1), pred.view(bs, n_p, -1), obj.view(1, R, C)

    def _refine(self, roi_feats, boxes_m, prop_feats, bs, n_p, attended=None):
        """roi_feats (R,S,C); boxes_m (bs,n_p,D) with centres in metres; prop_feats (bs,n_p,C)-viewable or None.
        Inference on the GPU runs `_refine_hip`; the torch formulation below is the autograd (training) form and what
        the host-side fixture tests exercise."""
        if self._hip_eligible(roi_feats):
            return self._refine_hip(roi_feats, boxes_m, prop_feats, bs, n_p, attended)
        C = self.feat_channels_lidar
        R = bs * n_p
        if prop_feats is None:
            prop_feats = roi_feats.mean(dim=1)
        q = prop_feats.reshape(bs, n_p, C).permute(1, 0, 2)
        q = self.norm1_lidar(q + self.dropout1_lidar(self.self_attn_lidar(q, q, value=q)[0]))
        q = q.permute(1, 0, 2).reshape(R, C)
        roi_feats.srf_bin_major = True
        inter = self.inst_interact_lidar(q, roi_feats)
        obj = self.norm2_lidar(q + self.dropout2_lidar(inter))
        ffn = self.linear2_lidar(self.dropout_lidar(self.activation_lidar(self.linear1_lidar(obj))))
        obj = self.norm3_lidar(obj + self.dropout3_lidar(ffn))
        cls_f, reg_f = obj, obj
        for layer in self.cls_module_lidar:
            cls_f = layer(cls_f)
        for layer in self.reg_module_lidar:
            reg_f = layer(reg_f)
        logits = self.class_logits_lidar(cls_f)
        deltas = self.bboxes_delta_lidar(reg_f)
        pred = self.apply_deltas_lidar(deltas, boxes_m.reshape(R, -1))
        return logits.view(bs, n_p, -1), pred.view(bs, n_p, -1), obj.view(1, R, C)

    def apply_deltas_lidar(self, deltas, boxes):
        """deltas (N,D) applied to boxes (N,D) [cx,cy,cz (m), log w,l,h, sin, cos, (vx,vy)] -> boxes with centres
        normalised to [0,1] by the range; sin/cos/velocity are taken from the deltas (srfdet_head.py:1534-1625)."""
        boxes = boxes.to(deltas.dtype)
        w = _const(self.bbox_weights[:6], deltas)
        d = deltas[:, :6] / w
        size = boxes[:, 3:6].exp()
        ctr = d[:, :3] * size + boxes[:, :3]
        new_size = d[:, 3:6].clamp(max=self.scale_clamp).exp() * size
        r = self.pc_range_lidar
        lo = _const(r[:3], deltas)
        ext = _const([r[3] - r[0], r[4] - r[1], r[5] - r[2]], deltas)
        ctr = ((ctr - lo) / ext).clamp(min=0.0, max=1.0)
        return torch.cat([ctr, new_size.log(), deltas[:, 6:]], dim=-1)


@HEADS.register_module()
class SingleSRFDetHeadLiDAR(_StageBase):
    def forward(self, point_feats, bboxes, prop_feats, pooler, img_metas=None):
        """(bs,n_p,D) boxes with normalised centres -> (logits (bs,n_p,#cls), boxes (bs,n_p,D), obj (1,bs*n_p,C)).
        `bboxes[..., :3]` is overwritten with metres, as in the reference."""
        bs, n_p = bboxes.shape[:2]

        def gather():
            boxes_m, rois, _ = self._geometry(bboxes, True, None)
            return self._gather(point_feats, rois, pooler), boxes_m
        roi_feats, boxes_m, attended = self._fork_gather(prop_feats, bs, n_p, gather, False)
        return self._refine(roi_feats, boxes_m, prop_feats, bs, n_p, attended)


@HEADS.register_module()
class SingleSRFDetHead(_StageBase):
    def __init__(self, *args, use_fusion=False, **kwargs):
        super().__init__(*args, use_fusion=use_fusion, **kwargs)

    def _fused_gather(self, img_feats, point_feats, rois_img, rois_bev, pooler_img, pooler, bs, n_p, n_cam):
        """Both gathers of a fusion stage straight into the operand of `output_fused_proj` (srfdet_head.py:2543-2566 gathers the
        n_cam * R image RoIs, sums the cameras, gathers the BEV RoIs and concatenates): the image gather sums the cameras itself
        (`srf_roi_extract_sum`) into the left half of a (R, S, 2 C) buffer, the BEV gather writes the right half -- two launches
        instead of four (gather, camera sum, gather, cat) and no (n_cam R, S, C) intermediate.  Inference on the GPU only."""
        C = self.feat_channels_lidar
        R = bs * n_p
        layer = pooler.roi_layers[0]
        S = layer.output_size ** 2
        fused_in = torch.empty((R, S, 2 * C), dtype=torch.float32, device=rois_bev.device)
        flat = [f.reshape(f.shape[0] * f.shape[1], *f.shape[2:]) for f in img_feats]
        if self.corrected_cam_indexing and bs > 1:
            ids = rois_img[:, 0]
            rois_img = rois_img.clone()
            rois_img[:, 0] = torch.remainder(ids, bs) * n_cam + torch.div(ids, bs, rounding_mode="floor")
        pooler_img(_channels_last(flat[:pooler_img.num_inputs]), rois_img, out=fused_in[..., :C], bin_major=True, n_sum=n_cam)
        pooler(_channels_last(list(point_feats[:pooler.num_inputs])), rois_bev, out=fused_in[..., C:], bin_major=True)
        return fused_in

    def forward(self, img_feats, point_feats, bboxes, prop_feats, pooler, img_metas, pooler_img=None):
        bs, n_p = bboxes.shape[:2]
        l2i = self._lidar2img(img_metas, bboxes) if img_feats is not None else None
        if (img_feats is not None and point_feats is not None and self.use_fusion and bboxes.is_cuda and not torch.is_grad_enabled()
                and isinstance(pooler, SingleRoIExtractor) and isinstance(pooler_img, SingleRoIExtractor)
                and self._hip_eligible(bboxes.new_empty((1, pooler.roi_layers[0].output_size ** 2, 1)))):
            def gather():
                boxes_m, rois_bev, rois_img = self._geometry(bboxes, True, l2i)
                fused_in = self._fused_gather(img_feats, point_feats, rois_img, rois_bev, pooler_img, pooler, bs, n_p, l2i.shape[1])
                return ops.linear(fused_in.view(-1, fused_in.shape[-1]), self.output_fused_proj.weight,
                                  self.output_fused_proj.bias).view(fused_in.shape[0], fused_in.shape[1], -1), boxes_m
            roi_feats, boxes_m, attended = self._fork_gather(prop_feats, bs, n_p, gather, True)
            return self._refine(roi_feats, boxes_m, prop_feats, bs, n_p, attended)
        bboxes, rois_bev, rois_img = self._geometry(bboxes, point_feats is not None, l2i)
        img_roi = self._img_rois_feats(img_feats, rois_img, pooler_img, bs, n_p, l2i.shape[1]) \
            if img_feats is not None else None
        pts_roi = self._gather(point_feats, rois_bev, pooler) if point_feats is not None else None
        if img_roi is not None and pts_roi is not None and self.use_fusion:
            fused_in = torch.cat((img_roi, pts_roi), dim=-1)
            if self._hip_eligible(pts_roi):
                roi_feats = ops.linear(fused_in.view(-1, fused_in.shape[-1]), self.output_fused_proj.weight,
                                       self.output_fused_proj.bias).view(pts_roi.shape)
            else:
                roi_feats = self.output_fused_proj(fused_in)
        elif not self.use_fusion and img_roi is not None and pts_roi is None:
            roi_feats = img_roi
        elif not self.use_fusion and pts_roi is not None and img_roi is None:
            roi_feats = pts_roi
        else:
            raise ValueError("unsupported combination of modalities / use_fusion")
        return self._refine(roi_feats, bboxes, prop_feats, bs, n_p)


@HEADS.register_module()
class SingleSRFDetHeadImg(_StageBase):
    """Registered because the configs carry a `single_head_img` dict; never built into the model
    (its construction is commented out upstream, srfdet_head.py:159-173)."""


@HEADS.register_module()
class SRFDetHead(BaseModule):
    def __init__(self, use_img=False, num_classes=4, feat_channels_lidar=256, feat_channels_img=256, hidden_dim=128,
                 lidar_feat_lvls=4, img_feat_lvls=4, num_proposals=128, num_heads=6, deep_supervision=True,
                 prior_prob=0.01, is_kitti=False, with_lidar_encoder=False, grid_size=None, out_size_factor=8,
                 lidar_encoder_cfg=None, code_weights=None, with_dpg=True, num_dpg_exp=4, single_head_lidar=None,
                 single_head_img=None, roi_extractor_lidar=None, roi_extractor_img=None, sync_cls_avg_factor=True,
                 loss_cls=None, loss_bbox=None, train_cfg=None, test_cfg=None, init_cfg=None, pretrained=None):
        super().__init__(init_cfg)
        assert not with_lidar_encoder, "with_lidar_encoder=False in every SRFDet3D config (SURVEY.md finding 8)"
        self.num_classes = num_classes
        self.use_img = use_img
        self.feat_channels_lidar = feat_channels_lidar
        self.feat_channels_img = feat_channels_img
        self.hidden_dim = hidden_dim
        self.lidar_feat_lvls = lidar_feat_lvls
        self.img_feat_lvls = img_feat_lvls
        self.num_proposals = num_proposals
        self.num_heads = num_heads
        self.deep_supervision = deep_supervision
        self.prior_prob = prior_prob
        self.is_kitti = is_kitti
        self.pc_range = single_head_lidar["pc_range"]
        self.test_cfg = test_cfg
        self.train_cfg = train_cfg
        self.with_dpg = with_dpg
        self.num_dpg_exp = num_dpg_exp
        self.grid_size = grid_size
        self.out_size_factor = out_size_factor
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.with_lidar_encoder = False
        self.use_fed_loss, self.use_focal_loss = False, True
        self._code_size = len(code_weights)

        if with_dpg:
            self._build_dynamic_prop_gen()
        else:
            self.init_proposal_boxes = nn.Embedding(num_proposals, self._code_size)
            self.init_proposal_feats = nn.Embedding(num_proposals, feat_channels_lidar)

        stage_cfg = dict(copy.deepcopy(single_head_lidar))
        stage_cfg.update(num_classes=num_classes, feat_channels=feat_channels_lidar,
                         pooler_resolution=roi_extractor_lidar["roi_layer"].get("output_size"),
                         use_focal_loss=self.use_focal_loss, use_fed_loss=self.use_fed_loss, is_kitti=is_kitti)
        stage = build_head(stage_cfg)
        self.head_series_lidar = ModuleList([copy.deepcopy(stage) for _ in range(num_heads)])
        self.roi_extractor_lidar = build_roi_extractor(roi_extractor_lidar)
        if use_img:
            if hidden_dim != feat_channels_img:
                self.img_convs = nn.ModuleList([
                    build_conv_layer(dict(type="Conv2d"), feat_channels_img, hidden_dim, kernel_size=3, padding=1,
                                     bias="auto") for _ in range(img_feat_lvls)])
            self.roi_extractor_img = build_roi_extractor(roi_extractor_img)
        else:
            self.roi_extractor_img = None

        self.loss_cls = LOSSES.build(loss_cls) if loss_cls and loss_cls["type"] in LOSSES else None
        self.loss_bbox = LOSSES.build(loss_bbox) if loss_bbox and loss_bbox["type"] in LOSSES else None
        self.assigner = None
        if train_cfg and train_cfg.get("assigner") and train_cfg["assigner"]["type"] in BBOX_ASSIGNERS:
            self.assigner_type = train_cfg["assigner"]["type"]
            self.assigner = BBOX_ASSIGNERS.build(train_cfg["assigner"])
        self.code_weights = nn.Parameter(torch.tensor(code_weights, requires_grad=False), requires_grad=False)
        self.use_nms = (test_cfg or {}).get("use_nms", True)
        self._init_weights()

    # ---- construction helpers -------------------------------------------------------------------------------
    def _dw_stair(self, channels, levels):
        return ModuleList([ConvModule(channels * (l + 1), channels * (l + 1), kernel_size=3, stride=2, padding=1,
                                      groups=channels * (l + 1), norm_cfg=dict(type="BN2d", eps=1e-3, momentum=0.01))
                           for l in range(levels - 1)])

    def _build_dynamic_prop_gen(self):
        n = self.num_dpg_exp * self.num_proposals
        self.init_proposal_boxes = nn.Embedding(n, self._code_size)
        self.init_proposal_feats = nn.Embedding(n, self.feat_channels_lidar)
        self.dpg_dw_convs_lidar = self._dw_stair(self.feat_channels_lidar, self.lidar_feat_lvls)
        down = self.out_size_factor * (2 ** (self.lidar_feat_lvls - 1))
        fx, fy = int(self.grid_size[0] / down), int(self.grid_size[1] / down)
        self.dpg_fc1_lidar = nn.Linear(fx * fy, 1024)
        self.dpg_act_lidar = nn.ReLU(inplace=True)
        self.dpg_fc2_lidar = nn.Linear(1024, n)
        if self.use_img:
            self.dpg_dw_convs_img = self._dw_stair(self.hidden_dim, self.img_feat_lvls)
            self.dpg_fc1_img = nn.Linear(30 * 15 if self.is_kitti else 30 * 30, 1500)
            self.dpg_act_img = nn.ReLU(inplace=True)
            self.dpg_fc2_img = nn.Linear(1500, n)

    def _init_weights(self):
        bias_value = -math.log((1 - self.prior_prob) / self.prior_prob)
        for name, p in self.named_parameters():
            if name in ("code_weights", "init_proposal_boxes.weight", "init_proposal_feats.weight"):
                continue
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
            if p.shape[-1] in (self.num_classes, self.num_classes + 1):
                nn.init.constant_(p, bias_value)

    # ---- proposals ------------------------------------------------------------------------------------------
    @staticmethod
    def _stair(convs, feats):
        """depthwise stride-2 stair over the pyramid with channel concat (srfdet_head.py:525-536).
        MIOpen resolves these depthwise convs to its naive reference kernel (34 ms on the 6 x 128 x 232 x 400 image
        level, measured), so it is kept away from the stair: in inference ConvModule hands them to srf_dwconv3x3s2
        (conv + BN + ReLU in one streaming kernel), otherwise torch's own depthwise kernel runs (0.2 ms)."""
        if SRFDetHead._stair_fusable(convs, feats):
            return SRFDetHead._stair_nhwc(convs, feats)
        with torch.backends.cudnn.flags(enabled=not feats[0].is_cuda):
            x = convs[0](feats[0])
            for lvl in range(1, len(feats)):
                x = torch.cat([feats[lvl], x], dim=1)
                if lvl < len(convs):
                    x = convs[lvl](x)
        return x

    @staticmethod
    def _dw_ok(cm):
        conv = cm.conv
        return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
                and conv.groups == conv.in_channels == conv.out_channels and conv.bias is None and cm.with_norm
                and _foldable(getattr(cm, cm.norm_name)) and cm.with_activation and isinstance(cm.activate, nn.ReLU))

    @staticmethod
    def _stair_nhwc(convs, feats, as_nhwc=False):
        """The same stair on channels-last levels: each depthwise conv + BN + ReLU writes its slice of the next level's
        concat buffer and copies the level itself into slice 0 in the same launch (torch.cat([feats[lvl], x], 1) of
        srfdet_head.py:531)."""
        x = nhwc.nhwc_view(feats[0])
        for lvl in range(1, len(feats)):
            f = nhwc.nhwc_view(feats[lvl])
            cm = convs[lvl - 1]
            c_lvl, c_x = f.shape[3], x.shape[3]
            buf = torch.empty((*f.shape[:3], c_lvl + c_x), dtype=torch.float32, device=f.device)
            scale, shift = _fold_bn2d(getattr(cm, cm.norm_name))
            x = ops.nhwc_dwconv3x3s2_cat(x, cm.conv.weight, scale, shift, True, f, buf)
            if lvl == len(feats) - 1 and lvl < len(convs):
                cm = convs[lvl]
                scale, shift = _fold_bn2d(getattr(cm, cm.norm_name))
                x = ops.nhwc_dwconv3x3s2(x, cm.conv.weight, scale, shift, True)
        return x if as_nhwc else nhwc.nchw_view(x)

    @staticmethod
    def _stair_fusable(convs, feats):
        return (nhwc.enabled() and all(fusable(f) and nhwc.is_channels_last(f) and f.shape[1] % 4 == 0 for f in feats)
                and all(SRFDetHead._dw_ok(c) for c in convs))

    @staticmethod
    def _channel_sum(x):
        """(bs, C, H, W) -> (bs, H*W), summed over channels (srfdet_head.py:537).  On the GPU it is done as a row
        reduction: torch's strided `sum(dim=1)` can become a multi-block "global reduce" whose captured form failed the
        replay validation (graphs._validate caught it on the image DPG and on the KITTI / Waymo BEV sizes; the cause inside
        torch is not established -- memset nodes themselves do replay correctly)."""
        if x.is_cuda and not torch.is_grad_enabled():
            return x.flatten(2).transpose(1, 2).contiguous().sum(dim=-1)
        return x.sum(dim=1).flatten(1, 2)

    def dpg_lidar_logits(self, point_feats):
        """The LiDAR half of the dynamic proposal generator (srfdet_head.py:506-512 of `_get_init_proposals`): depthwise stair over the
        BEV pyramid -> channel sum -> fc1 -> ReLU -> fc2, (bs, E * P).  It needs no image feature, so the LC frame computes it in the
        BEV-half graph, beside the camera graph, and hands it to `forward` (`precomputed_dpg_lidar`) instead of running its
        small launches on the serial tail after the join.  Inference on channels-last GPU levels: 6 launches (three stair steps,
        ops.nhwc_pool_sum, two GEMVs)."""
        feats = point_feats[:self.lidar_feat_lvls]
        if self._stair_fusable(self.dpg_dw_convs_lidar, feats):
            w = ops.nhwc_pool_sum(self._stair_nhwc(self.dpg_dw_convs_lidar, feats, as_nhwc=True))
        else:
            w = self._channel_sum(self._stair(self.dpg_dw_convs_lidar, feats))
        return linear_graph_safe(self.dpg_fc2_lidar, linear_graph_safe(self.dpg_fc1_lidar, w, relu=True, act=self.dpg_act_lidar))

    def dpg_img_logits(self, img_feats, bs):
        """The camera half (srfdet_head.py:541-552): stair over the camera levels, nearest resize to 30 x 30 (30 x 15 on KITTI), sum over
        the cameras and the channels, fc1 -> ReLU -> fc2."""
        flat = [f.reshape(f.shape[0] * f.shape[1], *f.shape[2:]) for f in img_feats[:self.img_feat_lvls]]
        n_cam = img_feats[0].shape[1]
        size = [30, 15] if self.is_kitti else [30, 30]
        if self._stair_fusable(self.dpg_dw_convs_img, flat):
            x = ops.nhwc_pool_sum(self._stair_nhwc(self.dpg_dw_convs_img, flat, as_nhwc=True), n_cam=n_cam, size=size)
        else:
            x = F.interpolate(self._stair(self.dpg_dw_convs_img, flat), size)
            x = self._channel_sum(x.view(bs, n_cam, *x.shape[1:]).sum(dim=1))
        return linear_graph_safe(self.dpg_fc2_img, linear_graph_safe(self.dpg_fc1_img, x, relu=True, act=self.dpg_act_img))

    def _get_init_proposals(self, img_feats, point_feats, lidar_logits=None):
        bs = point_feats[0].shape[0]
        boxes_w, feats_w = self.init_proposal_boxes.weight, self.init_proposal_feats.weight
        if not self.with_dpg:
            return boxes_w[None].repeat(bs, 1, 1), feats_w[None].repeat(bs, 1, 1)
        E, P = self.num_dpg_exp, self.num_proposals
        w = (self.dpg_lidar_logits(point_feats) if lidar_logits is None else lidar_logits).reshape(bs, E, P)
        if self.use_img:
            w = (w + self.dpg_img_logits(img_feats, bs).reshape(bs, E, P)) / 2
        w = w.softmax(1).unsqueeze(-1)
        boxes = (w * boxes_w.view(1, E, P, -1)).sum(1)
        feats = (w * feats_w.view(1, E, P, -1)).sum(1)
        return boxes, feats

    def _stage_proposals(self, img_feats, point_feats, lidar_logits=None):
        """`_get_init_proposals` + the sigmoid `forward` applies to the centres (srfdet_head.py:957).  GPU inference with the proposal
        generator: softmax, both expert sums and the sigmoid are one launch (ops.dpg_mix)."""
        if self.with_dpg and fusable(point_feats[0]):
            bs = point_feats[0].shape[0]
            wl = self.dpg_lidar_logits(point_feats) if lidar_logits is None else lidar_logits
            wi = self.dpg_img_logits(img_feats, bs) if self.use_img else None
            return ops.dpg_mix(wl.reshape(bs, -1), wi, self.init_proposal_boxes.weight, self.init_proposal_feats.weight,
                               self.num_dpg_exp, self.num_proposals)
        boxes, prop_feats = self._get_init_proposals(img_feats, point_feats, lidar_logits)
        boxes = boxes.contiguous()
        boxes[..., :3] = boxes[..., :3].sigmoid()
        return boxes, prop_feats

    def _img_convs_only(self, img_feats):
        """`img_convs` (3x3, feat_channels_img -> hidden_dim, bias) on every camera level (srfdet_head.py:404-416)."""
        out = list(img_feats)
        for i, f in enumerate(out):
            bs, n_cam, C, H, W = f.shape
            f4 = f.reshape(bs * n_cam, C, H, W)
            conv = self.img_convs[i]
            if nhwc.enabled() and fusable(f4) and nhwc.is_channels_last(f4) and nhwc.wino_ok(conv, C):
                g = nhwc.nchw_view(nhwc.conv3x3(nhwc.nhwc_view(f4), conv))  # Winograd on the f32 MFMA, bias in the epilogue
            else:
                from .. import train_conv
                g = train_conv.conv2d(conv, f4)     # training: srf_wino43 forward + data gradient; else conv(f4)
            out[i] = g.reshape(bs, n_cam, *g.shape[1:])
        return out

    def img_level_consumer(self):
        """`img_convs` as a per-level consumer for nhwc.level_consumer (channels-last in, channels-last out), or None when the
        head has no `img_convs` or a level cannot run on the Winograd kernel."""
        if not (self.use_img and self.hidden_dim != self.feat_channels_img and nhwc.enabled()):
            return None
        if not all(nhwc.wino_ok(conv, self.feat_channels_img) for conv in self.img_convs):
            return None
        return lambda i, x: nhwc.conv3x3(x, self.img_convs[i])

    # ---- forward --------------------------------------------------------------------------------------------
    def forward(self, img_feats, point_feats, img_metas, precomputed_dpg_lidar=None):
        """-> logits (#stage, bs, n_p, #cls), boxes (#stage, bs, n_p, D) with centres in metres, log sizes.
        precomputed_dpg_lidar: `dpg_lidar_logits(point_feats)` computed earlier by the caller (graphs.GraphedFrame: in the BEV half)."""
        logits_all, boxes_all = self._stages(img_feats, point_feats, img_metas, precomputed_dpg_lidar)
        r = self.pc_range
        lo = _const(r[:3], point_feats[0])
        ext = _const([r[3] - r[0], r[4] - r[1], r[5] - r[2]], point_feats[0])
        if self.deep_supervision:
            logits_all, boxes_all = torch.stack(logits_all), torch.stack(boxes_all)
        else:
            logits_all, boxes_all = logits_all[-1][None], boxes_all[-1][None].clone()
        boxes_all[..., :3] = boxes_all[..., :3] * ext + lo
        return logits_all, boxes_all

    def forward_decode(self, img_feats, point_feats, img_metas, precomputed_dpg_lidar=None):
        """`decode(*forward(...))` for inference: (scores (bs, n_p, #cls), boxes (bs, n_p, 7|9)).  On the GPU without autograd only the
        last stage's outputs are kept -- each stage rewrites the centres of its input boxes in metres in place (srfdet_head.py:1646),
        so the per-stage copies `forward` makes for deep supervision are not needed -- and the end of `forward` + `decode` run as one
        launch (ops.decode_boxes) instead of 13."""
        if not (point_feats[0].is_cuda and not torch.is_grad_enabled()):
            return self.decode(*self(img_feats, point_feats, img_metas, precomputed_dpg_lidar))
        logits, pred = self._stages(img_feats, point_feats, img_metas, precomputed_dpg_lidar, last_only=True)
        return ops.decode_boxes(logits[-1], pred[-1], self.pc_range)

    def _stages(self, img_feats, point_feats, img_metas, precomputed_dpg_lidar=None, last_only=False):
        """The decoder stages (srfdet_head.py:960-1000) -> per-stage lists of logits and boxes (centres normalised); with last_only
        one entry each, and no copy of the boxes between stages."""
        point_feats = list(point_feats)
        if self.use_img and self.hidden_dim != self.feat_channels_img and not isinstance(img_feats, nhwc.ConsumedLevels):
            img_feats = self._img_convs_only(img_feats)
        boxes, prop_feats = self._stage_proposals(img_feats, point_feats, precomputed_dpg_lidar)

        # channels-last copies of the pyramids, made once: every RoI tap then reads C contiguous floats
        if point_feats[0].is_cuda:
            point_feats_g = _channels_last(point_feats)
            img_feats_g = None
            if self.use_img:
                img_feats_g = [f.reshape(f.shape[0] * f.shape[1], *f.shape[2:]) for f in img_feats]
                img_feats_g = [f.view(img_feats[i].shape[0], img_feats[i].shape[1], *f.shape[1:])
                               for i, f in enumerate(_channels_last(img_feats_g))]
        else:
            point_feats_g, img_feats_g = point_feats, img_feats

        logits_all, boxes_all = [], []
        for stage in self.head_series_lidar:
            if not self.use_img:
                logits, pred, prop_feats = stage(point_feats_g, boxes, prop_feats, self.roi_extractor_lidar, img_metas)
            else:
                logits, pred, prop_feats = stage(img_feats_g, point_feats_g, boxes, prop_feats,
                                                 self.roi_extractor_lidar, img_metas, pooler_img=self.roi_extractor_img)
            if self.deep_supervision and not last_only:
                logits_all.append(logits)
                boxes_all.append(pred)
            # the next stage overwrites the centres of its input in place: a copy, unless this stage's boxes are not kept
            boxes = pred.detach() if last_only else pred.detach().clone()
        if last_only or not self.deep_supervision:
            logits_all, boxes_all = [logits], [pred]
        return logits_all, boxes_all

    # ---- training (srfdet_head.py:322-377, :1041-1201) ----------------------------------------------------------
    def forward_train(self, img_feats, point_feats, gt_bboxes, gt_labels, gt_bboxes_ignore=None, img_metas=None,
                      proposal_cfg=None, **kwargs):
        assert proposal_cfg is None and self.assigner is not None and self.loss_cls is not None
        logits, boxes = self(img_feats, point_feats, img_metas)
        out = dict(pred_logits=logits[-1], pred_boxes=boxes[-1])
        if self.deep_supervision:
            out["aux_outputs"] = [dict(pred_logits=a, pred_boxes=b) for a, b in zip(logits[:-1], boxes[:-1])]
        return self.loss_ota(out, gt_bboxes, gt_labels)

    def loss_ota(self, outputs, gt_bboxes_list, gt_labels_list):
        """srfdet_head.py:1041-1201.  The reference calls `reduce_mean` twice per decoder layer (the matched-pair count of the
        classification and of the box loss: 10 blocking all-reduces + `.item()` per iteration, :1134, :1178).  Both counts of a layer are
        the same number and every one of them is known once the assignments are made, so here the layers are assigned first and
        ALL counts travel in ONE all-reduce with ONE read-back (collective C4)."""
        from .training import reduce_mean
        dev = gt_labels_list[0].device
        gts = [torch.cat((g.gravity_center, g.tensor[:, 3:]), dim=1).to(dev) for g in gt_bboxes_list]
        losses = {}
        layers = [(outputs, self.num_heads, "")] + [(aux, i + 1, f"s.{i}.") for i, aux in enumerate(outputs.get("aux_outputs", []))]
        assigned = [self.assigner(out, gts, gt_labels_list, head_idx) for out, head_idx, _ in layers]
        local = [float(sum(int(j.numel()) for _, j in idx)) for idx in assigned]          # host-known: no synchronisation
        reduced = reduce_mean(torch.tensor(local, dtype=torch.float32, device=dev)).tolist()   # one collective, one read-back
        for (out, _, prefix), idx, n_local, n_mean in zip(layers, assigned, local, reduced):
            n_cls = n_mean if self.sync_cls_avg_factor else n_local
            losses[prefix + "loss_cls"] = self.loss_classification(out, gt_labels_list, idx, num=max(n_cls, 1.0))
            losses[prefix + "loss_bbox"] = self.loss_boxes(out, gts, idx, num=max(n_mean, 1.0))
        return losses

    def loss_classification(self, outputs, gt_labels_list, indices, num=None):
        from .training import reduce_mean
        logits = outputs["pred_logits"]
        target = torch.full(logits.shape[:2], self.num_classes, dtype=torch.int64, device=logits.device)
        n = 0
        for b, (labels, (fg, j)) in enumerate(zip(gt_labels_list, indices)):
            target[b, fg] = labels[j]
            n += int(j.numel())
        if num is None:   # stand-alone call: the reference's own sequence (one all-reduce for this layer)
            t = logits.new_tensor([float(n)])
            if self.sync_cls_avg_factor:
                t = reduce_mean(t)
            num = t.clamp(min=1).item()
        loss = self.loss_cls(logits.flatten(0, 1), target.flatten(0, 1)) / num
        return torch.nan_to_num(loss)

    def loss_boxes(self, outputs, gt_bboxes_list, indices, num=None):
        from .bbox_util import normalize_bbox
        from .training import reduce_mean
        pred = outputs["pred_boxes"]
        p = torch.cat([pred[b, fg] for b, (fg, _) in enumerate(indices)])
        t = torch.cat([g[j] for g, (_, j) in zip(gt_bboxes_list, indices)])
        if len(p) == 0:
            return torch.nan_to_num(pred.sum() * 0)
        if num is None:
            num = torch.clamp(reduce_mean(p.new_tensor([float(p.shape[0])])), min=1).item()
        tn = normalize_bbox(t, self.pc_range)
        ok = torch.isfinite(tn).all(dim=-1)
        D = self.code_weights.numel()
        w = torch.ones_like(p) * self.code_weights
        return torch.nan_to_num(self.loss_bbox(p[ok, :D], tn[ok, :D], w[ok, :D]) / num)

    def simple_test_bboxes(self, img_feats, point_feats, img_metas):
        return self.get_bboxes(None, None, img_metas, decoded=self.forward_decode(img_feats, point_feats, img_metas))

    def decode(self, pred_logits, pred_bboxes):
        """last-stage outputs -> (scores (bs,n_p,#cls), boxes (bs,n_p,7|9) with bottom-centre z): the tensors the
        reference hands to box3d_multiclass_nms (srfdet_head.py:1246-1271); the parity contract is on these."""
        scores = torch.sigmoid(pred_logits[-1])
        boxes = denormalize_bbox(pred_bboxes[-1], self.pc_range)
        boxes[..., 2] = boxes[..., 2] - boxes[..., 5] * 0.5
        return scores, boxes

    def select_static(self, scores, boxes):
        """Fixed-shape, sync-free form of the NMS of `get_bboxes` (for hipGraph capture): (bs,n_p,#cls), (bs,n_p,D) ->
        packed (bs, L, D+2) rows [box, score, label] whose first counts[b,0] rows are the survivors in the reference's order,
        and counts (bs, 2) int32 [survivors, candidates above the threshold]."""
        from ..postprocess import box3d_multiclass_nms_static
        cfg = self.test_cfg
        packed, counts = [], []
        for i in range(scores.shape[0]):
            p, c = box3d_multiclass_nms_static(boxes[i], scores[i], cfg["score_thr"], cfg["nms_thr"], want_packed=True)
            packed.append(p)
            counts.append(c)
        if len(packed) == 1:
            return packed[0].unsqueeze(0), counts[0].unsqueeze(0)
        return torch.stack(packed), torch.stack(counts)

    def results_from_static(self, packed, counts, img_metas):
        """Host side of `select_static`: packed / counts already on the CPU -> the list `get_bboxes` returns, or None when a
        sample had more candidates than the static capacity (the caller then runs `get_bboxes`).  numpy on the host rows: this
        sits between two frames on the critical path (a dozen small torch CPU ops cost ~100 us)."""
        import numpy as np
        cfg = self.test_cfg
        results = []
        pk = packed.numpy() if isinstance(packed, torch.Tensor) else packed
        cn = counts.numpy() if isinstance(counts, torch.Tensor) else counts
        L, D = pk.shape[1], pk.shape[2] - 2
        lo = np.asarray(cfg["post_center_range"][:3], dtype=pk.dtype)
        hi = np.asarray(cfg["post_center_range"][3:], dtype=pk.dtype)
        for i in range(pk.shape[0]):
            kept, cand = int(cn[i, 0]), int(cn[i, 1])
            if cand > L:
                return None
            rows = pk[i, :kept]
            if kept > cfg["max_per_img"]:
                rows = rows[np.argsort(-rows[:, D], kind="stable")[:cfg["max_per_img"]]]
            ctr = rows[:, :3]
            keep = ((ctr >= lo) & (ctr <= hi)).all(1)
            rows = np.ascontiguousarray(rows[keep])          # a copy: the pinned read-back buffer is reused by the next frame
            boxes = torch.from_numpy(rows[:, :D].copy())
            scores = torch.from_numpy(rows[:, D].copy())
            labels = torch.from_numpy(rows[:, D + 1].astype(np.int64))
            results.append([img_metas[i]["box_type_3d"](boxes, boxes.shape[-1]), scores, labels])
        return results

    def get_bboxes(self, pred_logits, pred_bboxes, img_metas, decoded=None):
        from ..postprocess import box3d_multiclass_nms
        cfg = self.test_cfg
        scores_all, boxes_all = decoded if decoded is not None else self.decode(pred_logits, pred_bboxes)
        results = []
        for i in range(scores_all.shape[0]):
            scores, boxes = scores_all[i], boxes_all[i]
            box_type = img_metas[i]["box_type_3d"]
            if self.use_nms:
                boxes, scores, labels = box3d_multiclass_nms(boxes, scores, cfg["score_thr"], cfg["max_per_img"],
                                                             cfg["nms_thr"])
            else:
                scores, idx = scores.flatten(0, 1).topk(cfg["max_per_img"])
                labels = idx % self.num_classes
                boxes = boxes[idx // self.num_classes]
            rng = _const(cfg["post_center_range"], scores)
            keep = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
            results.append([box_type(boxes[keep], boxes.shape[-1]), scores[keep], labels[keep]])
        return results
