"""The sliver of mmdet3d.core the decode path touches (srfdet_head.py:1276-1335, srfdet.py:332-340)."""
import torch


class LiDARInstance3DBoxes:
    """Holds (n, box_dim) boxes [x, y, z(bottom), dx, dy, dz, yaw, ...]; only what get_bboxes uses."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape(0, box_dim)
        self.tensor = tensor
        self.box_dim = box_dim
        self.with_yaw = with_yaw

    @property
    def bev(self):
        return self.tensor[:, [0, 1, 3, 4, 6]]

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = c[:, 2] + self.tensor[:, 5] * 0.5
        return c

    def to(self, device):
        return LiDARInstance3DBoxes(self.tensor.to(device), self.box_dim, self.with_yaw)

    def __len__(self):
        return self.tensor.shape[0]


def xywhr2xyxyr(b):
    """(cx, cy, w, h, r) -> (x1, y1, x2, y2, r)."""
    out = torch.zeros_like(b)
    hw, hh = b[..., 2] / 2, b[..., 3] / 2
    out[..., 0] = b[..., 0] - hw
    out[..., 1] = b[..., 1] - hh
    out[..., 2] = b[..., 0] + hw
    out[..., 3] = b[..., 1] + hh
    out[..., 4] = b[..., 4]
    return out


def bbox3d2result(bboxes, scores, labels):
    return dict(boxes_3d=bboxes.to("cpu"), scores_3d=scores.cpu(), labels_3d=labels.cpu())
