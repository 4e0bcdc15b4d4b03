"""Decode-side post-processing: per-class rotated NMS.

Mirrors mmdet3d `box3d_multiclass_nms` as called at mmdet3d_plugin/models/sparse_heads/srfdet_head.py:1276-1293:
per class keep score > score_thr, rotated NMS on the BEV boxes, concatenate, then top `max_num` by score.
All classes go through ONE NMS launch that carries the class id of every box and suppresses inside a class only
(`srf_nms_rotated_classes`), which gives the same keeps as the reference's per-class Python loop without moving any
coordinate.
"""
import torch

from . import ops


def box3d_multiclass_nms(boxes, scores, score_thr, max_num, nms_thr):
    """boxes (n, 7|9) [x,y,z,dx,dy,dz,yaw,...]; scores (n, #cls) -> (boxes, scores, labels), class-major order as
    the reference produces before its final top-k."""
    n, num_classes = scores.shape
    cand = (scores > score_thr).nonzero(as_tuple=False)  # (m, 2): box index, class -- row-major in box index
    if cand.shape[0] == 0:
        return boxes.new_zeros((0, boxes.shape[1])), scores.new_zeros((0,)), scores.new_zeros((0,), dtype=torch.long)
    if cand.shape[0] > NMS_MAX_BOXES:
        # more (box, class) pairs than one srf_nms_rotated launch takes (np = 900 x 10 classes with an untrained model):
        # run the classes one by one like the reference's loop -- a class never has more than n boxes
        return _per_class_nms(boxes, scores, score_thr, max_num, nms_thr)
    # class-major order, original box order inside a class (what the per-class loop yields)
    order = torch.argsort(cand[:, 1] * n + cand[:, 0])
    bi, ci = cand[order, 0], cand[order, 1]
    s = scores[bi, ci]
    bev = boxes[bi][:, [0, 1, 3, 4, 6]].contiguous()
    keep = ops.nms_rotated(bev, s, nms_thr, classes=ci)  # suppression inside a class only: the reference's per-class loop
    keep = keep.sort()[0]  # back to class-major / in-class score order is not needed before the top-k below
    # the reference appends, per class, boxes in descending score order
    k2 = torch.argsort(ci[keep] * 4 - s[keep].clamp(0, 1) * 2, stable=True)
    keep = keep[k2]
    out_b, out_s, out_l = boxes[bi[keep]], s[keep], ci[keep]
    if out_b.shape[0] > max_num:
        top = out_s.sort(descending=True)[1][:max_num]
        out_b, out_s, out_l = out_b[top], out_s[top], out_l[top]
    return out_b, out_s, out_l


NMS_MAX_BOXES = 4096  # boxes per srf_nms_rotated launch (64 mask words per row)


def _per_class_nms(boxes, scores, score_thr, max_num, nms_thr):
    out_b, out_s, out_l = [], [], []
    for c in range(scores.shape[1]):
        idx = (scores[:, c] > score_thr).nonzero(as_tuple=False).squeeze(1)
        if idx.numel() == 0:
            continue
        if idx.numel() > NMS_MAX_BOXES:
            # more candidates of ONE class than a launch takes (64 mask words per row): truncating to the best 4096 would differ
            # from the reference's loop once suppression removes many of them, so the greedy NMS runs exactly, in blocks of
            # descending score -- a block is first thinned by the survivors so far, then by itself
            keep = _nms_rotated_blocks(boxes[idx][:, [0, 1, 3, 4, 6]].contiguous(), scores[idx, c], nms_thr)
            keep = idx[keep]
        else:
            keep = idx[ops.nms_rotated(boxes[idx][:, [0, 1, 3, 4, 6]].contiguous(), scores[idx, c], nms_thr)]  # descending score
        out_b.append(boxes[keep])
        out_s.append(scores[keep, c])
        out_l.append(torch.full((keep.numel(),), c, dtype=torch.long, device=boxes.device))
    if not out_b:
        return boxes.new_zeros((0, boxes.shape[1])), scores.new_zeros((0,)), scores.new_zeros((0,), dtype=torch.long)
    out_b, out_s, out_l = torch.cat(out_b), torch.cat(out_s), torch.cat(out_l)
    if out_b.shape[0] > max_num:
        top = out_s.sort(descending=True)[1][:max_num]
        out_b, out_s, out_l = out_b[top], out_s[top], out_l[top]
    return out_b, out_s, out_l


def _nms_rotated_blocks(bev, scores, thr, block=NMS_MAX_BOXES):
    """Exact greedy rotated NMS of any number of boxes of one class on launches of at most NMS_MAX_BOXES boxes; returns the kept
    indices in descending score.  Blocks of descending score: a block is first thinned by the survivors of the earlier blocks
    (`ops.box_iou_rotated` of survivors x block: only KEPT boxes may suppress), then by itself (`ops.nms_rotated`)."""
    order = scores.argsort(descending=True, stable=True)
    kept = order.new_zeros((0,))
    for b0 in range(0, order.numel(), block):
        cand = order[b0:b0 + block]
        if kept.numel():
            dead = torch.zeros(cand.numel(), dtype=torch.bool, device=bev.device)
            for k0 in range(0, kept.numel(), 8192):
                iou = ops.box_iou_rotated(bev[kept[k0:k0 + 8192]].contiguous(), bev[cand].contiguous())
                dead |= (iou > thr).any(dim=0)
            cand = cand[~dead]
        if cand.numel():
            cand = cand[ops.nms_rotated(bev[cand].contiguous(), scores[cand], thr)]
        kept = torch.cat([kept, cand])
    return kept


STATIC_CANDIDATES = 2048  # capacity of the fixed-shape NMS (the kernel takes up to 4096)


def box3d_multiclass_nms_static(boxes, scores, score_thr, nms_thr, capacity=STATIC_CANDIDATES, want_packed=False):
    """The same selection with FIXED shapes and no host read-back, so that it can live inside a hipGraph.

    boxes (n, D), scores (n, C) -> (out_boxes (L, D), out_scores (L,), out_labels (L,), kept, candidates) with
    L = min(n*C, capacity): the first `kept` (device int32) rows are the NMS survivors in the reference's order (class-major,
    descending score inside a class); `candidates` (device int32) is the number of (box, class) pairs above the score
    threshold -- if it exceeds L the caller must redo the frame with `box3d_multiclass_nms`."""
    n, C = scores.shape
    L = min(n * C, int(capacity))
    if 0 < n * C <= 16384 and L <= 4096 and boxes.shape[1] >= 7:
        # two single-workgroup launches around the NMS (LDS bitonic sorts) instead of the ~25 small torch launches below
        cand, top_s, ci, bev, m = ops.nms_select(boxes, scores, score_thr, capacity)
        keep = ops.nms_rotated_counted(bev, m, nms_thr, classes=ci)  # the kernels take min(*m, L) themselves
        out_b, out_s, out_l, kept, packed, counts = ops.nms_finish(cand, top_s, ci, keep, m)
        if want_packed:
            return packed, counts
        return out_b, out_s, out_l, kept, m
    out = _static_torch(boxes, scores, score_thr, nms_thr, L)
    if want_packed:  # (L, D+2) rows [box, score, label] and [kept, candidates]: what select_static ships to the host
        b, s_, l, kept, m = out
        return torch.cat([b, s_.unsqueeze(1), l.to(b.dtype).unsqueeze(1)], dim=1), torch.cat([kept, m])
    return out


def _static_torch(boxes, scores, score_thr, nms_thr, L):
    """the same selection on torch ops (shapes beyond the fused kernels' limits; the definition the fused path is tested
    against)"""
    n, C = scores.shape
    flat = scores.reshape(-1)
    valid = flat > score_thr
    key = torch.where(valid, flat, torch.full_like(flat, -1.0))
    top_s, top_i = torch.topk(key, L, sorted=True)           # descending score, candidates first
    m = valid.sum().to(torch.int32).view(1)
    bi, ci = torch.div(top_i, C, rounding_mode="floor"), top_i % C
    cand = boxes[bi]
    # (a list index would upload an index tensor: not allowed while a stream is capturing)
    bev = torch.stack([cand[:, 0], cand[:, 1], cand[:, 3], cand[:, 4], cand[:, 6]], dim=1)
    keep = ops.nms_rotated_counted(bev, torch.clamp(m, max=L), nms_thr, classes=ci.contiguous()).bool()
    # survivors first, class-major, descending score inside a class (the per-class loop of the reference)
    key2 = torch.where(keep, ci.to(flat.dtype) * 4 - top_s.clamp(0, 1) * 2, torch.full_like(top_s, 1.0e9))
    o2 = torch.argsort(key2, stable=True)
    kept = keep.sum().to(torch.int32).view(1)
    return cand[o2], top_s[o2], ci[o2], kept, m
