"""The hot path end to end on the CPU: oracle operators + torch-CPU dense layers.

TEST INFRASTRUCTURE ONLY: the integration parity tests compare the HIP path against this on the same seeded frame,
and bench.py times it as `cpu_baseline` (kind "port").  It walks a CPU copy of the SAME torch modules the product
uses (so weights are shared) but calls oracle/oracle.py wherever the product calls a HIP kernel.

Reference path being restated: SRFDet.extract_point_features (mmdet3d_plugin/models/detectors/srfdet.py:249-276),
SparseEncoderCustom.forward (middle_encoders/sparse_encoder_custom.py:109-140), SRFDetHead.forward
(sparse_heads/srfdet_head.py:379-504).
"""
import numpy as np
import torch
from torch import nn

from . import decoder_oracle as DO
from . import oracle as O


def _np(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------- sparse encoder
class _Level:
    def __init__(self, feats, idx, shape):
        self.f, self.idx, self.shape = feats, idx, list(shape)
        self.subm = {}


def _fold(bn):
    return O.bn_fold(_np(bn.weight), _np(bn.bias), _np(bn.running_mean), _np(bn.running_var), bn.eps)


def _conv(conv, x, bn=None, relu=False, residual=None, stats=None):
    K = int(np.prod(conv.kernel_size))
    W = _np(conv.weight).reshape(K, conv.in_channels, conv.out_channels)
    if conv.subm:
        key = tuple(conv.kernel_size)
        if key not in x.subm:
            x.subm[key] = O.rulebook_subm(x.idx, x.shape, conv.kernel_size)
        nbr, counts = x.subm[key]
        out = _Level(None, x.idx, x.shape)
        out.subm = x.subm
    else:
        oi, nbr, counts, oshape = O.rulebook_strided(x.idx, x.shape, conv.kernel_size, conv.stride, conv.padding)
        out = _Level(None, oi, oshape)
    alpha, beta = _fold(bn) if bn is not None else (None, None)
    out.f = O.spconv_fwd(x.f, W, nbr, alpha, beta, residual, relu)
    if stats is not None:
        stats.append(dict(cin=conv.in_channels, cout=conv.out_channels, K=K, A_in=len(x.idx), A_out=len(out.idx),
                          pairs=int(counts.sum()), subm=bool(conv.subm)))
    return out


def _run_sequential(seq, x, stats):
    from srfdet3d_amd.sparse import SparseBasicBlock, SparseSequential, _SparseConv
    mods = list(seq._modules.values())
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, _SparseConv):
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
            relu = bn is not None and i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = _conv(m, x, bn, relu, None, stats)
            i += 1 + (bn is not None) + relu
        elif isinstance(m, SparseBasicBlock):
            mid = _conv(m.conv1, x, m.bn1, True, None, stats)
            x = _conv(m.conv2, mid, m.bn2, True, x.f, stats)
            i += 1
        elif isinstance(m, SparseSequential):
            x = _run_sequential(m, x, stats)
            i += 1
        else:
            raise TypeError(type(m))
    return x


def sparse_encoder(enc, voxel_feats, coors, batch_size, stats=None):
    """numpy (M,C), (M,4) -> numpy (B, C*D, H, W)."""
    x = _Level(np.ascontiguousarray(voxel_feats, np.float32), np.ascontiguousarray(coors, np.int32), enc.sparse_shape)
    x = _run_sequential(enc.conv_input, x, stats)
    x = _run_sequential(enc.encoder_layers, x, stats)
    x = _run_sequential(enc.conv_out, x, stats)
    dense = O.densify(x.f, x.idx, batch_size, x.shape)
    B, C, D, H, W = dense.shape
    return dense.reshape(B, C * D, H, W)


# ---------------------------------------------------------------------------------------------- point branch
def voxel_features(model, points_list):
    """list of numpy (N,C) -> (voxel feats (M,C'), coors (M,4)) following SRFDet.voxelize + the voxel encoder."""
    cfg = model.pts_voxel_layer
    if cfg.max_num_points != -1:
        feats, coors = [], []
        for b, pts in enumerate(points_list):
            v, c, n = O.hard_voxelize(pts, cfg.voxel_size, cfg.point_cloud_range, cfg.max_num_points, cfg.max_voxels[1])
            feats.append(O.vfe_mean(v, n, model.pts_voxel_encoder.num_features))
            coors.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
        return np.concatenate(feats, 0), np.concatenate(coors, 0)
    # dynamic voxelization + DynamicVFECustom (voxel_encoder.py:162-240) with the oracle scatter
    enc = model.pts_voxel_encoder
    pts = np.concatenate(points_list, 0).astype(np.float32)
    coors = np.concatenate([
        np.concatenate([np.full((len(p), 1), b, np.int32), O.dynamic_voxelize(p, cfg.voxel_size, cfg.point_cloud_range)], 1)
        for b, p in enumerate(points_list)], 0)
    g = O.grid_size(cfg.voxel_size, cfg.point_cloud_range)
    grid_zyx = [int(g[2]), int(g[1]), int(g[0])]
    parts = [pts]
    if enc._with_cluster_center:
        vmean, _, p2v = O.dynamic_scatter(pts, coors, grid_zyx, "mean")
        f_cluster = pts[:, :3] - vmean[np.maximum(p2v, 0)][:, :3]
        if enc._with_centroid_aware_vox:
            with torch.no_grad():
                f_cluster = enc.cen2point_pos_enc(torch.from_numpy(f_cluster)).numpy()
        parts.append(f_cluster)
    if enc._with_voxel_center:
        cf = coors.astype(np.float32)
        parts.append(np.stack([pts[:, 0] - (cf[:, 3] * np.float32(enc.vx) + np.float32(enc.x_offset)),
                               pts[:, 1] - (cf[:, 2] * np.float32(enc.vy) + np.float32(enc.y_offset)),
                               pts[:, 2] - (cf[:, 1] * np.float32(enc.vz) + np.float32(enc.z_offset))], 1))
    if enc._with_distance:
        parts.append(np.linalg.norm(pts[:, :3], axis=1, keepdims=True))
    x = np.concatenate(parts, 1).astype(np.float32)
    mode = "mean" if enc.vfe_scatter.average_points else "max"
    for i, vfe in enumerate(enc.vfe_layers):
        with torch.no_grad():
            pf = vfe(torch.from_numpy(x)).numpy()
        vf, vc, p2v = O.dynamic_scatter(pf, coors, grid_zyx, mode)
        if i != len(enc.vfe_layers) - 1:
            x = np.concatenate([pf, vf[np.maximum(p2v, 0)]], 1)
    return vf, vc


def point_features(model, points_list, stats=None):
    """-> tuple of torch CPU feature maps (the FPN outputs)."""
    vf, coors = voxel_features(model, points_list)
    bev = sparse_encoder(model.pts_middle_encoder, vf, coors, len(points_list), stats)
    with torch.no_grad():
        x = model.pts_backbone(torch.from_numpy(bev))
        if model.pts_neck is not None:
            x = model.pts_neck(x)
    return x


# ---------------------------------------------------------------------------------------------- decoder
def head_forward(head, img_feats, point_feats, img_metas, capture=None, stage_inputs=None):
    """SRFDetHead.forward on the CPU: stage geometry by decoder_oracle, gather by the oracle RoIAlign, stage
    arithmetic by the (reference-pinned, device-agnostic) torch code of the stage modules."""
    with torch.no_grad():
        point_feats = list(point_feats)
        if head.use_img and head.hidden_dim != head.feat_channels_img:
            img_feats = list(img_feats)
            for i, f in enumerate(img_feats):
                bs, n_cam = f.shape[:2]
                g = head.img_convs[i](f.reshape(bs * n_cam, *f.shape[2:]))
                img_feats[i] = g.reshape(bs, n_cam, *g.shape[1:])
        boxes, prop = head._get_init_proposals(img_feats, point_feats)
        boxes = boxes.clone()
        boxes[..., :3] = boxes[..., :3].sigmoid()
        pf_np = [_np(f) for f in point_feats[:head.roi_extractor_lidar.num_inputs]]
        if head.use_img:
            if_np = [_np(f.reshape(f.shape[0] * f.shape[1], *f.shape[2:])) for f in img_feats[:head.roi_extractor_img.num_inputs]]
            l2i = np.asarray([m["lidar2img"] for m in img_metas], np.float32)
            if l2i.ndim == 3:
                l2i = l2i[:, None]
        logits_all, boxes_all = [], []
        for si, stage in enumerate(head.head_series_lidar):
            if stage_inputs is not None:  # stage-by-stage comparison: feed the recorded inputs of this stage
                boxes, prop = (torch.as_tensor(t).clone() for t in stage_inputs[si])
            bs, P = boxes.shape[:2]
            b_np = _np(boxes)
            rois, bm = DO.lidar_rois(b_np, stage.pc_range_lidar, stage.voxel_size_lidar)
            roi, _ = O.roi_extract(pf_np, rois, head.roi_extractor_lidar.featmap_strides)
            roi = torch.from_numpy(roi).flatten(2).permute(0, 2, 1).contiguous()
            if head.use_img:
                rimg = DO.image_rois(b_np, stage.pc_range_lidar, l2i)
                ri, _ = O.roi_extract(if_np, rimg, head.roi_extractor_img.featmap_strides)
                ri = torch.from_numpy(ri).flatten(2).permute(0, 2, 1)
                ri = ri.reshape(l2i.shape[1], bs * P, ri.shape[1], ri.shape[2]).sum(0)
                roi = stage.output_fused_proj(torch.cat((ri, roi), dim=-1))
            if capture is not None:
                capture.append(dict(rois=rois, roi_feats=roi.numpy().copy()))
            logits, pred, prop = stage._refine(roi, torch.from_numpy(bm), prop, bs, P)
            logits_all.append(logits)
            boxes_all.append(pred)
            boxes = pred.clone()
        r = head.pc_range
        lo = torch.tensor(r[:3], dtype=torch.float32)
        ext = torch.tensor([r[3] - r[0], r[4] - r[1], r[5] - r[2]], dtype=torch.float32)
        logits_all, boxes_all = torch.stack(logits_all), torch.stack(boxes_all)
        boxes_all[..., :3] = boxes_all[..., :3] * ext + lo
    return logits_all, boxes_all


def forward_to_decode(model, points_list, img_metas, img=None, stats=None):
    """full CPU path up to the pre-NMS tensors: (scores (bs,P,#cls), boxes (bs,P,9))."""
    feats = point_features(model, points_list, stats)
    img_feats = None
    if img is not None:
        with torch.no_grad():
            img_feats = model.extract_img_feat(img, img_metas)
    logits, boxes = head_forward(model.bbox_head, img_feats, feats, img_metas)
    return model.bbox_head.decode(logits, boxes)
