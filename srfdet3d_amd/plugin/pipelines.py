"""The step before the path: the reference's `test_pipeline` between decoded sensor data and
`SRFDet.forward`, on the device.

Reference: configs/nus/srfdet_voxel_nusc_LC.py:253-283 (the pipeline), mmdet3d_plugin/datasets/pipelines/transform_3d.py:7-93
(`PadMultiViewImage`, `NormalizeMultiviewImage`), mmdet3d 1.0.0rc6 for `PointsRangeFilter`, `LoadPointsFromMultiSweeps`,
`MultiScaleFlipAug3D`, `DefaultFormatBundle3D`, `Collect3D` (third party: their semantics are restated here, parity unpinned).

What is different from the reference and why: the transforms keep their registry names, constructor arguments and the
`results` keys they read / write, but `results['points']` is a GPU (N, nf) float32 tensor and `results['img']` a GPU
(V, H, W, 3) uint8 tensor (the decoded views, as `LoadMultiViewImageFromFiles` stacks them) instead of numpy on the host:
the sweep of a frame is 0.6 MB and its six views 26 MB as bytes (107 MB once they are float32), so they are uploaded raw
and everything after the decode runs as two HBM-bound kernels (`ops.points_filter`, `ops.image_prepare`).  File reading
and image decoding (`LoadPointsFromFile`, `LoadMultiViewImageFromFiles`) stay outside: they are host I/O.
`Compose` fuses an adjacent Normalize + Pad pair into the one `image_prepare` launch.
"""
import numpy as np
import torch

from .. import ops
from ..compat.registry import PIPELINES


def _points_tensor(p):
    if not isinstance(p, torch.Tensor) or not p.is_cuda:
        raise RuntimeError("srfdet3d_amd: pipeline transforms take GPU tensors (no CPU fallback exists)")
    return p


@PIPELINES.register_module()
class PointsRangeFilter:
    """results['points'] <- the points strictly inside point_cloud_range, order kept (mmdet3d `points.in_range_3d`)."""

    def __init__(self, point_cloud_range):
        self.pcd_range = [float(v) for v in point_cloud_range]

    def __call__(self, results):
        results["points"] = ops.points_filter(_points_tensor(results["points"]), self.pcd_range)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(point_cloud_range={self.pcd_range})"


@PIPELINES.register_module()
class LoadPointsFromMultiSweeps:
    """The in-memory half of mmdet3d's transform of the same name: results['sweeps'] is a list of dicts with the already
    loaded 'points' (GPU (n, >= 5) tensor: x, y, z, intensity, ring/time slot), 'sensor2lidar_rotation' (3x3),
    'sensor2lidar_translation' (3), 'timestamp' (us); results['timestamp'] the key frame's in SECONDS (the dataset divides
    it, nuscenes_dataset.py:44).  Every sweep is cleaned of the
    returns from the ego vehicle (`remove_close`, radius 1 m), moved into the key frame's LiDAR frame, stamped with its age
    in seconds in channel 4, and appended to the key frame's points (whose channel 4 is set to 0)."""

    def __init__(self, sweeps_num=10, load_dim=5, use_dim=(0, 1, 2, 4), pad_empty_sweeps=False, remove_close=False,
                 test_mode=False, file_client_args=None):
        self.sweeps_num = sweeps_num
        self.load_dim = load_dim
        self.use_dim = list(use_dim)
        self.pad_empty_sweeps = pad_empty_sweeps
        self.remove_close = remove_close
        self.test_mode = test_mode

    def __call__(self, results):
        points = _points_tensor(results["points"]).clone()
        points[:, 4] = 0
        out = [points]
        ts = results["timestamp"]
        sweeps = results.get("sweeps", [])
        if self.pad_empty_sweeps and len(sweeps) == 0:
            for _ in range(self.sweeps_num):
                out.append(ops.points_filter(points, None, 1.0) if self.remove_close else points)
        else:
            if len(sweeps) <= self.sweeps_num or self.test_mode:
                choices = range(min(len(sweeps), self.sweeps_num))
            else:
                choices = np.random.choice(len(sweeps), self.sweeps_num, replace=False)
            for i in choices:
                sw = sweeps[int(i)]
                p = _points_tensor(sw["points"])[:, :self.load_dim]
                p = ops.points_filter(p, None, 1.0) if self.remove_close else p.clone()
                rot = torch.as_tensor(np.asarray(sw["sensor2lidar_rotation"], np.float32), device=p.device)
                p[:, :3] = p[:, :3] @ rot.T
                p[:, :3] += torch.as_tensor(np.asarray(sw["sensor2lidar_translation"], np.float32), device=p.device)
                p[:, 4] = ts - sw["timestamp"] / 1e6
                out.append(p)
        results["points"] = torch.cat(out, 0)[:, self.use_dim].contiguous()
        return results


@PIPELINES.register_module()
class NormalizeMultiviewImage:
    """(x - mean) / std per channel in float32, after a BGR -> RGB swap when to_rgb (transform_3d.py:59-93).  On its own it
    turns the (V, H, W, 3) uint8 views into (V, 3, H, W) float32; followed by PadMultiViewImage, `Compose` runs both as one
    launch."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb

    def _cfg(self, results):
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)

    def __call__(self, results, pad=None):
        img = results["img"]
        V, H, W, _ = img.shape
        if pad is None:
            size = (H, W)
        elif pad.size is not None:
            size = tuple(pad.size)
        else:
            d = pad.size_divisor
            size = (-(-H // d) * d, -(-W // d) * d)
        results["img"] = ops.image_prepare(img, self.mean, self.std, self.to_rgb, size=size)
        self._cfg(results)
        if pad is not None:
            pad._keys(results, (H, W), size)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})"


@PIPELINES.register_module()
class PadMultiViewImage:
    """Zero padding below / right of every view to a fixed size or to a multiple of size_divisor (transform_3d.py:7-56).
    Adds 'img_shape', 'pad_shape', 'pad_fixed_size', 'pad_size_divisor' like the reference (shapes as (H, W, 3) per view)."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert size is not None or size_divisor is not None
        assert size is None or size_divisor is None
        if pad_val != 0:
            raise NotImplementedError("srfdet3d_amd: PadMultiViewImage pads with zeros (every reference config does)")
        self.size = size
        self.size_divisor = size_divisor
        self.pad_val = pad_val

    def _keys(self, results, hw, size):
        V = results["img"].shape[0]
        results.setdefault("ori_shape", [(hw[0], hw[1], 3)] * V)
        results["img_shape"] = [(size[0], size[1], 3)] * V
        results["pad_shape"] = [(size[0], size[1], 3)] * V
        results["pad_fixed_size"] = self.size
        results["pad_size_divisor"] = self.size_divisor

    def __call__(self, results):
        img = results["img"]  # (V, 3, H, W) float32, already normalised
        if img.dtype == torch.uint8:
            raise RuntimeError("srfdet3d_amd: PadMultiViewImage follows NormalizeMultiviewImage in every reference pipeline")
        H, W = img.shape[-2:]
        if self.size is not None:
            size = tuple(self.size)
        else:
            d = self.size_divisor
            size = (-(-H // d) * d, -(-W // d) * d)
        results["img"] = torch.nn.functional.pad(img, (0, size[1] - W, 0, size[0] - H))
        self._keys(results, (H, W), size)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(size={self.size}, size_divisor={self.size_divisor}, pad_val={self.pad_val})"


@PIPELINES.register_module()
class DefaultFormatBundle3D:
    """The reference's bundle transposes every view HWC -> CHW, stacks them and wraps points / images as tensors.  Here the
    views already are one (V, 3, H, W) device tensor and the points a device tensor: nothing is left to do."""

    def __init__(self, class_names=None, with_gt=True, with_label=True):
        self.class_names = class_names

    def __call__(self, results):
        return results


@PIPELINES.register_module()
class Collect3D:
    """-> {'img_metas': dict of the meta keys present, key: results[key] for the requested keys}."""

    META_KEYS = ("filename", "ori_shape", "img_shape", "lidar2img", "depth2img", "cam2img", "pad_shape", "scale_factor", "flip",
                 "pcd_horizontal_flip", "pcd_vertical_flip", "box_mode_3d", "box_type_3d", "img_norm_cfg", "pcd_trans",
                 "sample_idx", "pcd_scale_factor", "pcd_rotation", "pcd_rotation_angle", "pts_filename", "transformation_3d_flow",
                 "trans_mat", "affine_aug")

    def __init__(self, keys, meta_keys=None):
        self.keys = list(keys)
        self.meta_keys = tuple(meta_keys) if meta_keys is not None else self.META_KEYS

    def __call__(self, results):
        data = {"img_metas": {k: results[k] for k in self.meta_keys if k in results}}
        for k in self.keys:
            data[k] = results[k]
        return data


@PIPELINES.register_module()
class MultiScaleFlipAug3D:
    """Test-time wrapper.  Every reference config uses one scale, pts_scale_ratio = 1 and flip = False, i.e. exactly one
    pass of the inner transforms whose outputs are wrapped in one-element lists; anything else is refused."""

    def __init__(self, transforms, img_scale=None, pts_scale_ratio=1, flip=False, flip_direction="horizontal",
                 pcd_horizontal_flip=False, pcd_vertical_flip=False):
        ratios = pts_scale_ratio if isinstance(pts_scale_ratio, (list, tuple)) else [pts_scale_ratio]
        if flip or pcd_horizontal_flip or pcd_vertical_flip or list(ratios) != [1]:
            raise NotImplementedError("srfdet3d_amd: test-time augmentation beyond the reference configs' single pass")
        self.transforms = Compose(transforms)
        self.img_scale = img_scale

    def __call__(self, results):
        results = dict(results)
        results.update(scale=self.img_scale, flip=False, pcd_scale_factor=1, pcd_horizontal_flip=False, pcd_vertical_flip=False)
        data = self.transforms(results)
        return {k: [v] for k, v in data.items()}


class Compose:
    """Builds the transforms of a config list through the PIPELINES registry and runs them in order; a
    NormalizeMultiviewImage directly followed by a PadMultiViewImage becomes one `image_prepare` launch."""

    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(dict(t)) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, results):
        i = 0
        while i < len(self.transforms):
            t = self.transforms[i]
            nxt = self.transforms[i + 1] if i + 1 < len(self.transforms) else None
            if isinstance(t, NormalizeMultiviewImage) and isinstance(nxt, PadMultiViewImage):
                results = t(results, pad=nxt)
                i += 2
            else:
                results = t(results)
                i += 1
            if results is None:
                return None
        return results


class FrameFeeder:
    """Host -> device staging of one frame's raw inputs (points float32, views uint8), double buffered: `put` of frame
    i+1 is issued before the compute of frame i is waited for; `get` makes the current stream wait for the copy (an
    event, no host sync) and hands out the device tensors.

    pinned=False (default): plain copies from pageable memory (the runtime stages them through its own bounce buffers).
    pinned=True: pinned staging buffers + non-blocking copies on a separate copy stream -- the textbook overlap, but on the
    MI355X boxes of this project every third copy out of a `pin_memory()` buffer stalled the device for 70-90 ms (blocking
    or not, with or without the copy stream, SDMA on or off; pageable copies never did), so it is opt-in."""

    def __init__(self, max_points, nf, views, height, width, device="cuda:0", depth=2, pinned=False):
        self.dev = torch.device(device)
        self.pinned = pinned
        self.stream = torch.cuda.Stream(self.dev) if pinned else None
        self.slots = []
        for _ in range(depth):
            s = dict(d_pts=torch.empty((max_points, nf), dtype=torch.float32, device=self.dev),
                     d_img=torch.empty((views, height, width, 3), dtype=torch.uint8, device=self.dev),
                     n=0, ready=torch.cuda.Event(), free=torch.cuda.Event())
            if pinned:
                s["h_pts"] = torch.empty((max_points, nf), dtype=torch.float32).pin_memory()
                s["h_img"] = torch.empty((views, height, width, 3), dtype=torch.uint8).pin_memory()
            self.slots.append(s)
        self._put = 0
        self._get = 0

    def put(self, points, images=None):
        s = self.slots[self._put % len(self.slots)]
        self._put += 1
        n = int(points.shape[0])
        s["n"] = n
        if not self.pinned:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(s["free"])  # the kernels that read this slot last are ahead of the copy on the device
            s["d_pts"][:n].copy_(torch.as_tensor(points))
            if images is not None:
                s["d_img"].copy_(torch.as_tensor(images))
            s["ready"].record(cur)
            return
        s["free"].synchronize()  # the compute that read this slot last has finished (no-op the first time round)
        s["h_pts"][:n].copy_(torch.as_tensor(points))
        if images is not None:
            s["h_img"].copy_(torch.as_tensor(images))
        with torch.cuda.stream(self.stream):
            s["d_pts"][:n].copy_(s["h_pts"][:n], non_blocking=True)
            if images is not None:
                s["d_img"].copy_(s["h_img"], non_blocking=True)
            s["ready"].record(self.stream)

    def get(self):
        s = self.slots[self._get % len(self.slots)]
        self._get += 1
        torch.cuda.current_stream(self.dev).wait_event(s["ready"])
        return s["d_pts"][:s["n"]], s["d_img"], s

    @staticmethod
    def release(slot):
        """call once the frame's kernels that read the slot are enqueued"""
        slot["free"].record(torch.cuda.current_stream())
