"""Training-side pieces of the reference that sit next to the hot path: losses, match costs, 3-D IoU,
the dynamic-k OTA label assignment and the frozen-LiDAR-branch helper.  Plain torch (label assignment is
O(np * n_gt)), except the rotated BEV IoU, which is a HIP kernel.

Mirrors: OTAssignerSRFDet  mmdet3d_plugin/core/bbox/assigners/ota_srfdet.py:18-327
         BBox3DL1Cost / IoU3DCost  mmdet3d_plugin/core/bbox/match_costs/match_cost.py:5-36
         FocalLoss / L1Loss / FocalLossCost / BboxOverlaps3D: mmdet 2.28 / mmdet3d 1.0rc6 semantics (not in the reference
         tree, "parity unpinned"); freeze_lidar_components  tools/train.py:221-276.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..compat.registry import BBOX_ASSIGNERS, LOSSES, MATCH_COST
from .bbox_util import boxes3d_to_corners3d, denormalize_bbox, normalize_bbox


def reduce_mean(t):
    """mmdet.core.utils.reduce_mean: mean over ranks (collective C4 of SURVEY.md 2.3)."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.clone()
    dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0, activated=False):
        super().__init__()
        assert use_sigmoid and not activated
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        """pred (N, C) logits; target (N,) labels with C = background."""
        C = pred.size(1)
        t = F.one_hot(target, num_classes=C + 1)[:, :C].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * fw
        if weight is not None:
            loss = loss * weight.view(-1, 1)
        if self.reduction == "sum":
            loss = loss.sum()
        elif self.reduction == "mean":
            loss = loss.sum() / avg_factor if avg_factor is not None else loss.mean()
        return self.loss_weight * loss


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        loss = (pred - target).abs()
        if weight is not None:
            loss = loss * weight
        if self.reduction == "sum":
            loss = loss.sum()
        elif self.reduction == "mean":
            loss = loss.sum() / avg_factor if avg_factor is not None else loss.mean()
        return self.loss_weight * loss


@MATCH_COST.register_module()
class FocalLossCost:
    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12, binary_input=False):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class BBox3DL1Cost:
    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MATCH_COST.register_module()
class IoU3DCost:
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, iou):
        return -iou * self.weight


def bbox_overlaps_3d(boxes1, boxes2):
    """mmdet3d BboxOverlaps3D(coordinate='lidar'), mode 'iou': boxes (n, >=7) [x, y, z, dx, dy, dz, yaw] with z taken
    as the bottom face (the reference hands gravity centres to it on both sides, ota_srfdet.py:148-150) -> (n, m)."""
    n, m = boxes1.shape[0], boxes2.shape[0]
    if n * m == 0:
        return boxes1.new_zeros((n, m))
    top1, top2 = boxes1[:, 2] + boxes1[:, 5], boxes2[:, 2] + boxes2[:, 5]
    h = (torch.min(top1.view(-1, 1), top2.view(1, -1)) - torch.max(boxes1[:, 2].view(-1, 1), boxes2[:, 2].view(1, -1))).clamp(min=0)
    bev1 = boxes1[:, [0, 1, 3, 4, 6]].clone()
    bev2 = boxes2[:, [0, 1, 3, 4, 6]].clone()
    bev1[:, 2:4] = bev1[:, 2:4].clamp(min=1e-4)
    bev2[:, 2:4] = bev2[:, 2:4].clamp(min=1e-4)
    iou2d = ops.box_iou_rotated(bev1.contiguous(), bev2.contiguous())
    a1 = (bev1[:, 2] * bev1[:, 3]).view(-1, 1)
    a2 = (bev2[:, 2] * bev2[:, 3]).view(1, -1)
    inter3d = iou2d * (a1 + a2) / (1 + iou2d) * h
    v1 = (boxes1[:, 3] * boxes1[:, 4] * boxes1[:, 5]).view(-1, 1)
    v2 = (boxes2[:, 3] * boxes2[:, 4] * boxes2[:, 5]).view(1, -1)
    return inter3d / (v1 + v2 - inter3d).clamp(min=1e-8)


@MATCH_COST.register_module()
class ClassificationCost:
    """mmdet's softmax classification cost (the default `cls_cost` of HungarianAssignerSRFDet)."""

    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, cls_pred, gt_labels):
        return -cls_pred.softmax(-1)[:, gt_labels] * self.weight


@MATCH_COST.register_module()
class BBoxL1Cost(BBox3DL1Cost):
    """mmdet's name for the plain L1 cost (default `reg_cost` of HungarianAssignerSRFDet); the box format conversions of
    the 2-D original do not apply to the normalised 3-D boxes it is given here."""

    def __init__(self, weight=1.0, box_format="xyxy"):
        super().__init__(weight)


class AssignResult:
    """The four fields of mmdet's AssignResult that callers of the assigner read."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


@BBOX_ASSIGNERS.register_module()
class HungarianAssignerSRFDet:
    """One-to-one matching of predictions to ground truth on cost = classification + L1 over the first 8 normalised box
    parameters, solved on the host with scipy (hungarian_assigner_srfdet.py:14-129).  No config of the reference enables
    it (the lines are commented out in favour of OTAssignerSRFDet); it is here so that those lines resolve."""

    def __init__(self, cls_cost=None, reg_cost=None, pc_range=None):
        self.cls_cost = MATCH_COST.build(cls_cost if cls_cost is not None else dict(type="ClassificationCost", weight=1.0))
        self.reg_cost = MATCH_COST.build(reg_cost if reg_cost is not None else dict(type="BBoxL1Cost", weight=1.0))
        self.pc_range = pc_range

    @torch.no_grad()
    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_bboxes_ignore=None, eps=1e-7, kdistillation=False):
        """-> AssignResult: gt_inds[i] = 0 for background or 1 + index of the matched ground truth; labels[i] = its class
        or -1."""
        from scipy.optimize import linear_sum_assignment
        from .bbox_util import normalize_bbox
        assert gt_bboxes_ignore is None, "gt_bboxes_ignore is not supported (as in the reference)"
        n_gt, n_q = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((n_q,), 0 if n_gt == 0 else -1, dtype=torch.long)
        labels = bbox_pred.new_full((n_q,), -1, dtype=torch.long)
        if n_gt == 0 or n_q == 0:
            return AssignResult(n_gt, gt_inds, None, labels=labels)
        target = gt_bboxes if kdistillation else normalize_bbox(gt_bboxes, self.pc_range)
        cost = self.cls_cost(cls_pred, gt_labels) + self.reg_cost(bbox_pred[:, :8], target[:, :8])
        rows, cols = linear_sum_assignment(cost.detach().cpu().numpy())
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        return AssignResult(n_gt, gt_inds, None, labels=labels)


@BBOX_ASSIGNERS.register_module()
class OTAssignerSRFDet(nn.Module):
    """1-to-k dynamic matching of predictions to ground truth (ota_srfdet.py:18-327)."""

    def __init__(self, cls_cost, reg_cost, iou_cost, center_radius=1.5, candidate_topk=5, pc_range=None, iou_calculator=None,
                 num_heads=6):
        super().__init__()
        self.center_radius, self.candidate_topk, self.pc_range, self.num_heads = center_radius, candidate_topk, pc_range, num_heads
        self.cls_cost, self.reg_cost, self.iou_cost = MATCH_COST.build(cls_cost), MATCH_COST.build(reg_cost), MATCH_COST.build(iou_cost)

    def forward(self, outputs, gt_boxes_list, gt_labels_list, head_idx):
        return [self.single_assigner(outputs["pred_boxes"][i], outputs["pred_logits"][i], gt_boxes_list[i], gt_labels_list[i],
                                     head_idx) for i in range(len(gt_boxes_list))]

    @torch.no_grad()
    def single_assigner(self, pred_boxes, pred_logits, gt_boxes, gt_labels, head_idx):
        n_gt = gt_boxes.size(0)
        if n_gt == 0:
            return pred_boxes.new_zeros((pred_boxes.shape[0],), dtype=torch.bool), pred_boxes.new_zeros((0,), dtype=torch.long)
        valid, in_both = self._in_gt_and_center(pred_boxes, gt_boxes)
        cls = self.cls_cost(pred_logits, gt_labels)
        reg = self.reg_cost(pred_boxes[:, :8], normalize_bbox(gt_boxes[:, :7], self.pc_range))
        ious = bbox_overlaps_3d(denormalize_bbox(pred_boxes, self.pc_range), gt_boxes)
        cost = cls + reg + self.iou_cost(ious) + (~in_both) * 100.0
        cost[~valid] = cost[~valid] + 10000.0
        return self._dynamic_k(cost, ious, n_gt, head_idx)

    def _in_gt_and_center(self, pred, gt):
        c = pred[:, None, :3]                                                  # (n_p, 1, 3)
        corners = boxes3d_to_corners3d(gt[None, :, :7], bottom_center=False, ry=True)[0]   # (n_gt, 8, 3)
        lo, hi = corners.min(dim=1).values[None], corners.max(dim=1).values[None]
        in_box = ((c > lo) & (c < hi)).all(-1)                                  # (n_p, n_gt)
        r = self.center_radius * gt[None, :, 3:6]
        in_ctr = ((c > gt[None, :, :3] - r) & (c < gt[None, :, :3] + r)).all(-1)
        return in_box.any(1) | in_ctr.any(1), in_box & in_ctr

    def _dynamic_k(self, cost, ious, n_gt, head_idx):
        topk = torch.topk(ious, min(self.candidate_topk, ious.size(0)), dim=0).values
        ks = torch.clamp((topk.sum(0) - 0.5 * (self.num_heads - head_idx)).int(), min=1)
        # the reference walks the ground-truth boxes: `topk(cost[:, g], k = ks[g].item(), largest=False)` per box (ota_srfdet.py:296-300) --
        # one read-back of ks and, per box, a top-k and an index_put launch (200 + 200 launches per step at 20 boxes x 10 assignments).
        # The same set in one pass: a prediction matches box g when its rank in the ascending order of cost[:, g] is below ks[g].
        order = torch.argsort(cost, dim=0, stable=True)
        rank = torch.empty_like(order)
        rank.scatter_(0, order, torch.arange(cost.shape[0], device=cost.device)[:, None].expand_as(order))
        match = (rank < ks[None, :]).to(cost.dtype)
        multi = match.sum(1) > 1
        if multi.any():
            best = cost[multi].argmin(dim=1)
            match[multi] = 0
            match[multi, best] = 1.0
        while (match.sum(0) == 0).any():
            cost[match.sum(1) > 0] += 100000.0
            for g in torch.nonzero(match.sum(0) == 0, as_tuple=False).squeeze(1).tolist():
                match[cost[:, g].argmin(), g] = 1.0
            if (match.sum(1) > 1).any():  # the reference re-uses the FIRST multi-match mask here (ota_srfdet.py:309-313)
                best = cost[multi].argmin(dim=1)
                match[multi] = match[multi] * 0
                match[multi, best] = 1.0
        fg = match.sum(1) > 0
        return fg, match[fg].argmax(1)


def freeze_lidar_components(model):
    """tools/train.py:221-276: stop gradients in every pts_* sub-module and keep their BatchNorm in eval."""
    for name in ("pts_voxel_encoder", "pts_middle_encoder", "pts_backbone", "pts_neck"):
        mod = getattr(model, name, None)
        if mod is None:
            continue
        mod.eval()
        for p in mod.parameters():
            p.requires_grad = False
        mod.train = lambda mode=True, _m=mod: nn.Module.train(_m, False)  # stays in eval under model.train()
    return model
