"""Box (de)normalisation and corner geometry, device-agnostic torch.

Mirrors mmdet3d_plugin/core/bbox/util.py: normalize_bbox (:4-38), denormalize_bbox (:41-81),
boxes3d_to_corners3d (:84-176) -- same argument meaning and output layout, without the hard-coded `.cuda()` calls
of :134,143-145.  The decoder's hot path does not call boxes3d_to_corners3d: its geometry is fused into
srf_box_rois (csrc/roi.hip); this torch form exists for callers outside the hot path and for the parity tests.
"""
import torch


def normalize_bbox(bboxes, pc_range=None):
    """(n,7|9) [x,y,z,w,l,h,rot,(vx,vy)] -> (n,8|10) [x,y,z,log w,log l,log h,sin,cos,(vx,vy)]."""
    rot = bboxes[..., 6:7]
    parts = [bboxes[..., 0:3], bboxes[..., 3:6].log(), rot.sin(), rot.cos()]
    if bboxes.size(-1) > 7:
        parts.append(bboxes[..., 7:9])
    return torch.cat(parts, dim=-1)


def denormalize_bbox(normalized_bboxes, pc_range=None):
    """(n,8|10) -> (n,7|9): sizes exp'ed, (sin,cos) -> yaw; centres are passed through."""
    rot = torch.atan2(normalized_bboxes[..., 6:7], normalized_bboxes[..., 7:8])
    parts = [normalized_bboxes[..., 0:3], normalized_bboxes[..., 3:6].exp(), rot]
    if normalized_bboxes.size(-1) > 8:
        parts.append(normalized_bboxes[..., 8:10])
    return torch.cat(parts, dim=-1)


_SX = (1., -1., -1., 1., 1., -1., -1., 1.)
_SY = (-1., -1., 1., 1., -1., -1., 1., 1.)


def boxes3d_to_corners3d(boxes3d, bottom_center=True, ry=False):
    """(bs,N,8) [cx,cy,cz,log w,log l,log h,sin,cos] (or (bs,N,7) with yaw if ry) -> (bs,N,8,3) corners."""
    if ry:
        yaw = boxes3d[..., 6]
    else:
        yaw = torch.atan2(boxes3d[..., 6], boxes3d[..., 7])
    w, l, h = boxes3d[..., 3].exp(), boxes3d[..., 4].exp(), boxes3d[..., 5].exp()
    sx = boxes3d.new_tensor(_SX)
    sy = boxes3d.new_tensor(_SY)
    xc = (w / 2.).unsqueeze(-1) * sx
    yc = (l / 2.).unsqueeze(-1) * sy
    if bottom_center:
        zc = h.unsqueeze(-1) * boxes3d.new_tensor((0., 0., 0., 0., 1., 1., 1., 1.))
    else:
        zc = (h / 2.).unsqueeze(-1) * boxes3d.new_tensor((-1., -1., -1., -1., 1., 1., 1., 1.))
    c, s = torch.cos(yaw).unsqueeze(-1), torch.sin(yaw).unsqueeze(-1)
    # row vector times [[c,-s,0],[s,c,0],[0,0,1]]
    x = boxes3d[..., 0:1] + (xc * c + yc * s)
    y = boxes3d[..., 1:2] + (xc * (-s) + yc * c)
    z = boxes3d[..., 2:3] + zc
    return torch.stack([x, y, z], dim=-1).float()
