"""The step before the path: PointsRangeFilter / remove_close and Normalize + Pad of the camera views.
CPU part: the oracle's restatement against the definitions it cites, on cases small enough to spell out; registry wiring.
GPU part (-m gpu): the HIP kernels against the oracle, bit for bit, and the reference's test_pipeline built through Compose."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from srfdet3d_amd import ops, synthetic as S
from srfdet3d_amd.compat.registry import PIPELINES

NUSC_LC_RANGE = [-55.2, -55.2, -5.0, 55.2, 55.2, 3.0]          # configs/nus/srfdet_voxel_nusc_LC.py:11
IMG_NORM = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)   # :15-18


def test_oracle_points_filter_known_answers():
    r = [0.0, 0.0, 0.0, 10.0, 10.0, 4.0]
    pts = np.array([[1, 1, 1, 7, 0],      # inside
                    [0, 5, 1, 7, 0],      # x == x_min: strict inequality drops it
                    [10, 5, 1, 7, 0],     # x == x_max: dropped
                    [5, 5, 4, 7, 0],      # z == z_max: dropped
                    [9.999, 9.999, 3.999, 7, 0],
                    [0.5, 0.5, 1, 7, 0],  # inside the range, but within 1 m of the sensor in x and y
                    [0.5, 3.0, 1, 7, 0],  # |x| < 1 but |y| >= 1: kept by remove_close
                    [np.nan, 1, 1, 7, 0]], np.float32)   # NaN compares false: dropped by the range test
    kept, idx = O.points_filter(pts, r)
    np.testing.assert_array_equal(idx, [0, 4, 5, 6])
    kept, idx = O.points_filter(pts, r, close_radius=1.0)
    np.testing.assert_array_equal(idx, [0, 4, 6])
    kept, idx = O.points_filter(pts, None, close_radius=1.0)       # remove_close alone keeps NaN x (|nan| < r is false)
    np.testing.assert_array_equal(idx, [0, 1, 2, 3, 4, 6, 7])
    np.testing.assert_array_equal(kept, pts[idx])
    kept, idx = O.points_filter(np.zeros((0, 5), np.float32), r)
    assert kept.shape == (0, 5) and idx.shape == (0,)


def test_oracle_image_prepare_known_answers():
    img = np.zeros((2, 3, 5, 3), np.uint8)
    img[0, 0, 0] = [10, 20, 30]
    img[1, 2, 4] = [255, 0, 128]
    out = O.image_prepare(img, [1.0, 2.0, 3.0], [2.0, 4.0, 8.0], to_rgb=False, size_divisor=4)
    assert out.shape == (2, 3, 4, 8) and out.dtype == np.float32
    np.testing.assert_array_equal(out[0, :, 0, 0], [(10 - 1) / 2, (20 - 2) / 4, (30 - 3) / 8])
    np.testing.assert_array_equal(out[1, :, 2, 4], [(255 - 1) / 2, (0 - 2) / 4, (128 - 3) / 8])
    assert np.all(out[:, :, 3:, :] == 0) and np.all(out[:, :, :, 5:] == 0)          # padding is zero, not (0 - mean) / std
    np.testing.assert_array_equal(out[0, :, 1, 1], [-0.5, -0.5, -0.375])            # a black pixel inside the image is not
    rgb = O.image_prepare(img, [1.0, 2.0, 3.0], [2.0, 4.0, 8.0], to_rgb=True, size_divisor=4)
    np.testing.assert_array_equal(rgb[0, :, 0, 0], [(30 - 1) / 2, (20 - 2) / 4, (10 - 3) / 8])   # channels swapped first
    fixed = O.image_prepare(img, [0, 0, 0], [1, 1, 1], size=(8, 8))
    assert fixed.shape == (2, 3, 8, 8)


def test_pipeline_registry_names_of_the_reference_test_pipeline():
    for name in ("PointsRangeFilter", "LoadPointsFromMultiSweeps", "NormalizeMultiviewImage", "PadMultiViewImage",
                 "MultiScaleFlipAug3D", "DefaultFormatBundle3D", "Collect3D"):
        assert PIPELINES.get(name) is not None, name
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.points_filter(torch.zeros(4, 5), NUSC_LC_RANGE)
    with pytest.raises(RuntimeError, match="GPU uint8"):
        ops.image_prepare(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), [0, 0, 0], [1, 1, 1])
    with pytest.raises(NotImplementedError):
        PIPELINES.build(dict(type="MultiScaleFlipAug3D", transforms=[], flip=True))


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 777, 30000, 300000])
def test_gpu_points_filter_bit_exact(dev, n):
    rng = np.random.default_rng(n)
    pts = np.concatenate([rng.uniform(-70, 70, (n, 2)), rng.uniform(-7, 5, (n, 1)), rng.uniform(0, 255, (n, 2))], 1).astype(np.float32)
    if n > 10:
        pts[3, 0] = NUSC_LC_RANGE[0]      # exactly on the faces
        pts[4, 1] = NUSC_LC_RANGE[4]
        pts[5, 2] = np.nan
    for rng_box, radius in ((NUSC_LC_RANGE, 0.0), (None, 1.0), (NUSC_LC_RANGE, 1.0)):
        want, widx = O.points_filter(pts, rng_box, radius)
        got, gidx = ops.points_filter(torch.from_numpy(pts).to(dev), rng_box, radius, with_index=True)
        np.testing.assert_array_equal(gidx.cpu().numpy(), widx)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        out, num = ops.points_filter(torch.from_numpy(pts).to(dev), rng_box, radius, static=True)
        assert int(num.item()) == len(widx)
        np.testing.assert_array_equal(out[:len(widx)].cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("V,H,W,div", [(6, 900, 1600, 32), (1, 370, 1224, 32), (5, 37, 50, 32), (2, 8, 8, 4)])
def test_gpu_image_prepare_bit_exact(dev, V, H, W, div):
    rng = np.random.default_rng(V * 1000 + H)
    img = rng.integers(0, 256, (V, H, W, 3), dtype=np.uint8)
    for to_rgb in (False, True):
        want = O.image_prepare(img, IMG_NORM["mean"], IMG_NORM["std"], to_rgb, div)
        got = ops.image_prepare(torch.from_numpy(img).to(dev), IMG_NORM["mean"], IMG_NORM["std"], to_rgb, div)
        assert tuple(got.shape) == want.shape
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_reference_test_pipeline_through_compose(dev):
    """the tail of configs/nus/srfdet_voxel_nusc_LC.py:253-283 from the decoded data on (file loading is host I/O)."""
    from srfdet3d_amd.plugin.pipelines import Compose
    pipe = Compose([
        dict(type="LoadPointsFromMultiSweeps", sweeps_num=10, use_dim=[0, 1, 2, 3, 4], test_mode=True),
        dict(type="NormalizeMultiviewImage", **IMG_NORM),
        dict(type="PadMultiViewImage", size_divisor=32),
        dict(type="MultiScaleFlipAug3D", img_scale=(1333, 800), pts_scale_ratio=1, flip=False, transforms=[
            dict(type="PointsRangeFilter", point_cloud_range=NUSC_LC_RANGE),
            dict(type="DefaultFormatBundle3D", class_names=["car"], with_label=False),
            dict(type="Collect3D", keys=["points", "img"])])])
    rng = np.random.default_rng(0)
    key = S.nuscenes_sweep(2000, 3000)
    sweeps, host_sweeps = [], []
    for i in range(2):
        p = S.nuscenes_sweep(2001 + i, 2000)
        a = 0.02 * (i + 1)
        rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        tr = np.array([0.5 * (i + 1), 0.1, 0.0], np.float32)
        host_sweeps.append((p, rot, tr, 1000000 - 50000 * (i + 1)))
        sweeps.append(dict(points=torch.from_numpy(p).to(dev), sensor2lidar_rotation=rot, sensor2lidar_translation=tr,
                           timestamp=1000000 - 50000 * (i + 1)))
    img = rng.integers(0, 256, (6, 90, 160, 3), dtype=np.uint8)
    data = pipe(dict(points=torch.from_numpy(key).to(dev), sweeps=sweeps, timestamp=1.0, img=torch.from_numpy(img).to(dev),
                     lidar2img=[np.eye(4, dtype=np.float32)] * 6))
    # the same on the host: numpy restatement of LoadPointsFromMultiSweeps (remove_close off, the config's default) + the oracle
    k = key.copy()
    k[:, 4] = 0
    parts = [k]
    for p, rot, tr, ts in host_sweeps:
        q = p.copy()
        q[:, :3] = q[:, :3] @ rot.T
        q[:, :3] += tr
        q[:, 4] = 1.0 - ts / 1e6
        parts.append(q)
    want_pts, _ = O.points_filter(np.concatenate(parts, 0), NUSC_LC_RANGE)
    got_pts = data["points"][0].cpu().numpy()
    assert got_pts.shape == want_pts.shape
    np.testing.assert_allclose(got_pts, want_pts, rtol=0, atol=2e-5)     # the 3x3 rotation runs on rocBLAS vs numpy
    np.testing.assert_array_equal(data["img"][0].cpu().numpy(), O.image_prepare(img, IMG_NORM["mean"], IMG_NORM["std"], False, 32))
    meta = data["img_metas"][0]
    assert meta["pad_shape"][0] == (96, 160, 3) and meta["ori_shape"][0] == (90, 160, 3) and len(meta["lidar2img"]) == 6
    assert meta["img_norm_cfg"]["to_rgb"] is False and meta["pcd_scale_factor"] == 1 and meta["flip"] is False


@pytest.mark.gpu
def test_gpu_frame_feeder_hands_over_what_was_put(dev):
    from srfdet3d_amd.plugin.pipelines import FrameFeeder
    rng = np.random.default_rng(1)
    feeder = FrameFeeder(5000, 5, 2, 16, 24, dev)
    frames = [(rng.standard_normal((4000 + 100 * i, 5)).astype(np.float32), rng.integers(0, 256, (2, 16, 24, 3), dtype=np.uint8))
              for i in range(5)]
    feeder.put(*frames[0])
    for i in range(5):
        pts, img, slot = feeder.get()
        if i + 1 < 5:
            feeder.put(*frames[i + 1])           # the next frame's upload is issued before this one is consumed
        got_p, got_i = pts.clone(), img.clone()
        FrameFeeder.release(slot)
        np.testing.assert_array_equal(got_p.cpu().numpy(), frames[i][0])
        np.testing.assert_array_equal(got_i.cpu().numpy(), frames[i][1])
