#!/usr/bin/env python3
"""Round-2 fixtures from the reference's own Python (run in the build container only): tests/golden/extra_r2.npz.

Same recipe as make_fixtures.py (reference files executed where they lie under /root/reference with name-only
stand-ins for mmcv / mmdet / mmdet3d; inputs and weights regenerated on both sides from detgen): only expected OUTPUTS
are stored.  New here:

* `vfe_kitti.*`, `vfe_waymo.*`   DynamicVFELayer + DynamicVFECustom.forward (voxel_encoders/utils.py:30-45,
                                 voxel_encoder.py:162-240).  Unpinned part: mmcv's DynamicScatter, restated below from
                                 SURVEY.md Appendix B.3 (sorted unique voxels, mean / max per voxel).
* `second.*`                     SECONDCustom.forward (backbones/second_custom.py:78-91) with the KITTI arguments.
* `kstage.*`                     one SingleSRFDetHeadLiDAR stage with the KITTI arguments (C = 256, d = 64, ff = 1024,
                                 3 classes, 8 box parameters) at P = 100.
* `lstage200.*`, `lstage900.*`   the nuScenes stage at P = 200 and P = 900.
* `ota.*`                        OTAssignerSRFDet.forward (core/bbox/assigners/ota_srfdet.py:57-327) and
                                 SRFDetHead.loss_ota / loss_classification / loss_boxes (srfdet_head.py:1042-1201).
                                 Unpinned parts (third-party, restated): mmdet's FocalLossCost / FocalLoss / L1Loss and
                                 mmdet3d's BboxOverlaps3D; the 3-D IoU matrices the reference run saw are stored so that
                                 the HIP IoU can be checked against them.

usage:  python tests/golden/make_fixtures_r2.py [--ref /root/reference]
"""
import argparse
import importlib.util
import os
import sys
import types
import zlib

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import detgen  # noqa: E402
import make_fixtures as mf  # noqa: E402

KITTI_RANGE = [0, -40, -3, 70.4, 40, 1]
KITTI_VOXEL = [0.05, 0.05, 0.1]
WAYMO_RANGE = [-76.8, -76.8, -2, 76.8, 76.8, 4]
WAYMO_VOXEL = [0.1, 0.1, 0.15]
KSTAGE_KW = dict(num_classes=3, feat_channels=256, pooler_resolution=7, dim_feedforward=1024, num_cls_convs=2,
                 num_reg_convs=3, num_heads=8, dropout=0.1, dynamic_conv=dict(dynamic_dim=64, dynamic_num=2),
                 bbox_weights=[1.0] * 8, pc_range=KITTI_RANGE, voxel_size=KITTI_VOXEL)
OTA_KW = dict(cls_cost=dict(type="FocalLossCost", weight=2.0, alpha=0.25, gamma=2.0, eps=1e-8),
              reg_cost=dict(type="BBox3DL1Cost", weight=0.25), iou_cost=dict(type="IoU3DCost", weight=0.25),
              center_radius=2.5, candidate_topk=8, pc_range=mf.NUSC_RANGE, num_heads=6)


# ------------------------------------------------------------------------------------------------
# shared deterministic inputs (the tests call these too)
# ------------------------------------------------------------------------------------------------
def vfe_points(name, pc_range, voxel_size, n_feat, n_vox=70, max_pts=9):
    """Points clustered into `n_vox` voxels (1..max_pts points each), all strictly inside the range, in a shuffled order;
    -> (points (N, n_feat) f32, coors (N, 4) int32 (b, z, y, x)) for a batch of 2 samples (sample id non-decreasing, as
    SRFDet.voxelize concatenates them)."""
    r = np.asarray(pc_range, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    grid = np.round((r[3:] - r[:3]) / vs).astype(np.int64)
    rng = np.random.default_rng(zlib.crc32(name.encode("utf-8")))
    pts, batch = [], []
    for b in range(2):
        cells = np.stack([rng.integers(2, grid[0] - 2, n_vox), rng.integers(2, grid[1] - 2, n_vox), rng.integers(1, grid[2] - 1, n_vox)], 1)
        rows = []
        for c in cells:
            k = int(rng.integers(1, max_pts + 1))
            xyz = r[:3] + (c[None, :] + rng.uniform(0.1, 0.9, (k, 3))) * vs
            rows.append(np.concatenate([xyz, rng.uniform(0, 1, (k, n_feat - 3))], 1))
        p = np.concatenate(rows, 0).astype(np.float32)
        p = p[rng.permutation(len(p))]
        pts.append(p)
        batch.append(np.full(len(p), b, np.int32))
    pts = np.concatenate(pts, 0)
    c = np.floor((pts[:, :3] - r[:3]) / vs).astype(np.int32)            # float32 arithmetic, as the voxelizer
    coors = np.stack([np.concatenate(batch), c[:, 2], c[:, 1], c[:, 0]], 1).astype(np.int32)
    return pts, coors


def ota_case(name, P, n_gt_list, n_cls=10, pc_range=mf.NUSC_RANGE):
    """Predictions placed around ground-truth boxes so that the dynamic-k matching has real work to do.
    -> pred_logits (bs, P, n_cls), pred_boxes (bs, P, 10) [normalised centres, log sizes, sin, cos, v], gt boxes (list of (n, 9)
    gravity-centre boxes [x y z dx dy dz yaw vx vy]), gt labels (list of (n,) int64)."""
    rng = np.random.default_rng(zlib.crc32(name.encode("utf-8")))
    r = np.asarray(pc_range, np.float64)
    bs = len(n_gt_list)
    logits = rng.standard_normal((bs, P, n_cls)).astype(np.float32)
    boxes = np.zeros((bs, P, 10), np.float32)
    gts, labels = [], []
    for b, n_gt in enumerate(n_gt_list):
        ctr = np.stack([rng.uniform(-40, 40, n_gt), rng.uniform(-40, 40, n_gt), rng.uniform(-2, 0.5, n_gt)], 1)
        size = np.stack([rng.uniform(1.5, 5.0, n_gt), rng.uniform(1.2, 2.5, n_gt), rng.uniform(1.2, 2.2, n_gt)], 1)
        yaw = rng.uniform(-np.pi, np.pi, (n_gt, 1))
        vel = rng.standard_normal((n_gt, 2))
        gts.append(np.concatenate([ctr, size, yaw, vel], 1).astype(np.float32))
        labels.append(rng.integers(0, n_cls, n_gt).astype(np.int64))
        for p in range(P):
            if n_gt and p < int(0.7 * P):          # near a ground-truth box
                g = p % n_gt
                c = ctr[g] + rng.standard_normal(3) * np.array([0.8, 0.8, 0.3])
                s = size[g] * np.exp(rng.standard_normal(3) * 0.15)
                y = yaw[g, 0] + rng.standard_normal() * 0.2
                logits[b, p, labels[b][g]] += 2.0
            else:                                   # background
                c = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-3, 1)])
                s = np.array([rng.uniform(1, 6), rng.uniform(1, 3), rng.uniform(1, 3)])
                y = rng.uniform(-np.pi, np.pi)
            boxes[b, p, :3] = (c - r[:3]) / (r[3:] - r[:3])
            boxes[b, p, 3:6] = np.log(s)
            boxes[b, p, 6], boxes[b, p, 7] = np.sin(y), np.cos(y)
            boxes[b, p, 8:] = rng.standard_normal(2) * 0.5
    return logits, boxes, gts, labels


# ------------------------------------------------------------------------------------------------
# restatements of the third-party pieces (the unpinned parts)
# ------------------------------------------------------------------------------------------------
class DynamicScatterRestated:
    """mmcv.ops.DynamicScatter: sorted unique (b, z, y, x) voxels; per-voxel mean or max."""

    def __init__(self, voxel_size, point_cloud_range, average_points):
        self.average_points = average_points

    def __call__(self, feats, coors):
        uniq, inv = torch.unique(coors.long(), dim=0, sorted=True, return_inverse=True)
        M, C = uniq.shape[0], feats.shape[1]
        if self.average_points:
            out = torch.zeros(M, C, dtype=feats.dtype).index_add_(0, inv, feats)
            cnt = torch.zeros(M, dtype=feats.dtype).index_add_(0, inv, torch.ones(len(inv), dtype=feats.dtype))
            out = out / cnt[:, None]
        else:
            out = torch.full((M, C), -float("inf"), dtype=feats.dtype).scatter_reduce(0, inv[:, None].expand(-1, C), feats, "amax")
        return out, uniq.to(coors.dtype)


def _poly_clip(subject, clip):
    def inside(p, a, b):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0

    def inter(p1, p2, a, b):
        d1, d2 = (p2[0] - p1[0], p2[1] - p1[1]), (b[0] - a[0], b[1] - a[1])
        den = d1[0] * d2[1] - d1[1] * d2[0]
        t = ((a[0] - p1[0]) * d2[1] - (a[1] - p1[1]) * d2[0]) / den
        return (p1[0] + t * d1[0], p1[1] + t * d1[1])

    out = subject
    for i in range(len(clip)):
        a, b = clip[i], clip[(i + 1) % len(clip)]
        inp, out = out, []
        if not inp:
            break
        s = inp[-1]
        for e in inp:
            if inside(e, a, b):
                if not inside(s, a, b):
                    out.append(inter(s, e, a, b))
                out.append(e)
            elif inside(s, a, b):
                out.append(inter(s, e, a, b))
            s = e
    return out


def _rect(x, y, dx, dy, yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    pts = [(-dx / 2, -dy / 2), (dx / 2, -dy / 2), (dx / 2, dy / 2), (-dx / 2, dy / 2)]
    return [(x + c * px - s * py, y + s * px + c * py) for px, py in pts]


def _area(poly):
    return 0.5 * abs(sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1] for i in range(len(poly))))


def bbox_overlaps_3d_f64(b1, b2):
    """mmdet3d BboxOverlaps3D(coordinate='lidar') on [x y z dx dy dz yaw ...] with z as the bottom face: rotated BEV
    intersection (float64 polygon clipping) x height overlap / union volume."""
    b1, b2 = b1.double().numpy(), b2.double().numpy()
    out = np.zeros((len(b1), len(b2)))
    for i, p in enumerate(b1):
        rp = _rect(p[0], p[1], p[3], p[4], p[6])
        for j, g in enumerate(b2):
            h = min(p[2] + p[5], g[2] + g[5]) - max(p[2], g[2])
            if h <= 0:
                continue
            poly = _poly_clip(rp, _rect(g[0], g[1], g[3], g[4], g[6]))
            inter = (_area(poly) if len(poly) >= 3 else 0.0) * h
            out[i, j] = inter / max(p[3] * p[4] * p[5] + g[3] * g[4] * g[5] - inter, 1e-8)
    return torch.from_numpy(out).float()


def install_extra_stand_ins(ref):
    from srfdet3d_amd.plugin import training as T  # this repo's restatements of the mmdet losses / focal cost (unpinned)

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def build_norm_layer(cfg, n):
        t = cfg["type"]
        cls = nn.BatchNorm1d if t in ("BN1d", "naiveSyncBN1dCustom") else nn.BatchNorm2d
        return "bn", cls(n, eps=cfg.get("eps", 1e-5), momentum=cfg.get("momentum", 0.1))

    def build_conv_layer(cfg, *a, **k):
        cfg = dict(cfg)
        assert cfg.pop("type") == "Conv2d"
        return nn.Conv2d(*a, **k, **cfg)

    mod("mmcv.cnn", build_norm_layer=build_norm_layer, build_conv_layer=build_conv_layer)
    # utils.py was executed by make_fixtures.load_reference with the BN1d-only stand-in bound to its global name
    sys.modules["mmdet3d_plugin.models.voxel_encoders.utils"].build_norm_layer = build_norm_layer
    mod("mmdet.models", BACKBONES=mf._Registry())
    mod("mmdet3d.ops", DynamicScatter=DynamicScatterRestated)
    mod("mmdet3d.models.builder", VOXEL_ENCODERS=mf._Registry(), build_fusion_layer=mf._raiser("build_fusion_layer"))
    mod("mmengine")
    mod("mmengine.structures", InstanceData=type("InstanceData", (), {}))
    mod("mmdet.core.bbox")
    mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=mf._Registry())
    mod("mmdet.core.bbox.match_costs")
    mod("mmdet.core.bbox.match_costs.builder", MATCH_COST=mf._Registry())
    spec = importlib.util.spec_from_file_location("mmdet3d_plugin.core.bbox.match_costs.match_cost",
                                                  os.path.join(ref, "mmdet3d_plugin/core/bbox/match_costs/match_cost.py"))
    mc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mc)

    def build_match_cost(cfg):
        cfg = dict(cfg)
        t = cfg.pop("type")
        if t == "FocalLossCost":
            return T.FocalLossCost(**cfg)       # mmdet's, restated (unpinned)
        return getattr(mc, t)(**cfg)             # the reference's own BBox3DL1Cost / IoU3DCost

    ious_seen = []

    def build_iou_calculator(cfg):
        def calc(a, b):
            iou = bbox_overlaps_3d_f64(a, b)
            ious_seen.append(iou.numpy().copy())
            return iou
        return calc

    mod("mmdet.core.bbox.match_costs", build_match_cost=build_match_cost)
    mod("mmdet.core.bbox.iou_calculators", build_iou_calculator=build_iou_calculator)
    return T, ious_seen


def load(ref, name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ref, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _GtBoxes:
    """The two members of LiDARInstance3DBoxes loss_ota reads (srfdet_head.py:1062-1065); built from gravity-centre rows."""

    def __init__(self, g):
        self.gravity_center = g[:, :3]
        self.tensor = g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    mf.install_stand_ins()
    util, head = mf.load_reference(args.ref)
    T, ious_seen = install_extra_stand_ins(args.ref)
    torch.manual_seed(0)
    t = torch.from_numpy
    out = {}

    # ---- DynamicVFECustom (KITTI: one layer 39 -> 4; Waymo: two layers 40 -> 5, 10 -> 5) ------------------------------
    ve = load(args.ref, "mmdet3d_plugin.models.voxel_encoders.voxel_encoder", "mmdet3d_plugin/models/voxel_encoders/voxel_encoder.py")
    for tag, kw, nf in (("vfe_kitti", dict(in_channels=4, feat_channels=[4], voxel_size=KITTI_VOXEL, point_cloud_range=KITTI_RANGE), 4),
                        ("vfe_waymo", dict(in_channels=5, feat_channels=[5, 5], voxel_size=WAYMO_VOXEL, point_cloud_range=WAYMO_RANGE), 5)):
        enc = ve.DynamicVFECustom(with_cluster_center=True, with_voxel_center=True, with_distance=False,
                                  norm_cfg=dict(type="naiveSyncBN1dCustom", eps=1e-3, momentum=0.01), **kw).eval()
        detgen.load_det_params(enc, tag + ".")
        pts, coors = vfe_points(tag, kw["point_cloud_range"], kw["voxel_size"], nf)
        with torch.no_grad():
            vf, vc = enc(t(pts), t(coors))
        out[tag + ".coors_in"] = coors
        out[tag + ".voxel_feats"], out[tag + ".voxel_coors"] = vf.numpy(), vc.numpy()

    # ---- SECONDCustom with the KITTI arguments on a 24 x 20 map -----------------------------------------------------
    sc = load(args.ref, "mmdet3d_plugin.models.backbones.second_custom", "mmdet3d_plugin/models/backbones/second_custom.py")
    net = sc.SECONDCustom(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2]).eval()
    detgen.load_det_params(net, "second.")
    with torch.no_grad():
        o = net(t(detgen.det("second.x", (1, 256, 24, 20), scale=0.5)))
    out["second.out0"], out["second.out1"] = o[0].numpy(), o[1].numpy()

    # ---- stages: KITTI arguments at P = 100; nuScenes arguments at P = 200 and P = 900 ---------------------------------
    for tag, kw, P, C, D in (("kstage", KSTAGE_KW, 100, 256, 8), ("lstage200", mf.STAGE_KW, 200, 128, 10),
                             ("lstage900", mf.STAGE_KW, 900, 128, 10)):
        st = head.SingleSRFDetHeadLiDAR(**kw).eval()
        detgen.load_det_params(st, tag + ".")
        bx = t(mf.det_boxes(tag + ".boxes", P)[..., :D].copy())
        pooler = mf.RecordingPooler(t(detgen.det(tag + ".roi_feats", (P, C, 7, 7))))
        with torch.no_grad():
            logits, pred, obj = st([None] * 4, bx, t(detgen.det(tag + ".prop", (1, P, C))), pooler, None)
        out[tag + ".rois"], out[tag + ".boxes_after"] = pooler.rois.numpy(), bx.numpy()
        out[tag + ".logits"], out[tag + ".pred"], out[tag + ".obj"] = logits.numpy(), pred.numpy(), obj.numpy()

    # ---- OTA assigner + loss_ota (two samples, one final + two auxiliary stages) -----------------------------------------
    ota = load(args.ref, "mmdet3d_plugin.core.bbox.assigners.ota_srfdet", "mmdet3d_plugin/core/bbox/assigners/ota_srfdet.py")
    assigner = ota.OTAssignerSRFDet(**OTA_KW)
    P = 64
    stages = [ota_case(f"ota.s{i}", P, [7, 5]) for i in range(3)]
    gts, labels = stages[0][2], stages[0][3]          # the same ground truth for every stage
    outputs = dict(pred_logits=t(stages[0][0]), pred_boxes=t(stages[0][1]),
                   aux_outputs=[dict(pred_logits=t(s[0]), pred_boxes=t(s[1])) for s in stages[1:]])
    for head_idx, o in ((6, outputs), (1, outputs["aux_outputs"][0]), (2, outputs["aux_outputs"][1])):
        del ious_seen[:]
        res = assigner(o, [t(g) for g in gts], [t(l) for l in labels], head_idx)
        for b, (fg, gi) in enumerate(res):
            out[f"ota.h{head_idx}.fg{b}"], out[f"ota.h{head_idx}.gt{b}"] = fg.numpy(), gi.numpy()
            out[f"ota.h{head_idx}.iou{b}"] = ious_seen[b]
    # an empty sample
    fg, gi = assigner.single_assigner(outputs["pred_boxes"][0], outputs["pred_logits"][0], torch.zeros(0, 9), torch.zeros(0, dtype=torch.long), 6)
    out["ota.empty.fg"], out["ota.empty.gt"] = fg.numpy(), gi.numpy()

    head.reduce_mean = lambda x: x  # one process
    hd = object.__new__(head.SRFDetHead)
    nn.Module.__init__(hd)
    hd.assigner, hd.num_heads, hd.deep_supervision, hd.num_classes, hd.sync_cls_avg_factor = assigner, 6, True, 10, True
    hd.pc_range = mf.NUSC_RANGE
    hd.code_weights = nn.Parameter(torch.tensor([1.0] * 8 + [0.2, 0.2]), requires_grad=False)
    hd.loss_cls = T.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="sum", loss_weight=2.0)   # mmdet's, restated
    hd.loss_bbox = T.L1Loss(reduction="sum", loss_weight=0.25)
    losses = hd.loss_ota(outputs, [_GtBoxes(t(g)) for g in gts], [t(l) for l in labels])
    for k, v in losses.items():
        out["ota.loss." + k] = np.asarray(float(v), np.float32)

    path = os.path.join(HERE, "extra_r2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if v.ndim else float(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
