"""Seeded synthetic sweeps for measurement and parity tests (numpy only).

The shapes follow SURVEY.md section 8(d): a spinning-LiDAR ring model inside the
point-cloud range of the unchanged reference configs
(`configs/nus/srfdet_voxel_nusc_L.py:11`, `configs/kitti/srfdet_voxel_kitti_L.py`,
`configs/waymo/srfdet_dvoxel_waymo_L.py:6-10`).  No dataset is available offline, so
every measured frame is generated here; `sha256_of` pins frame 0 of each workload.
"""
import hashlib

import numpy as np

NUSC_RANGE = (-55.2, -55.2, -5.0, 55.2, 55.2, 3.0)
KITTI_RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)
WAYMO_RANGE = (-76.8, -76.8, -2.0, 76.8, 76.8, 4.0)


def _ring_sweep(rng, n_points, n_rings, elev_lo_deg, elev_hi_deg, sensor_h,
                max_ground_range, up_range, az_lo, az_hi):
    """Ranges/angles of a spinning ring LiDAR over flat ground plus random up-beam returns.

    Points come in firing order, as in a real sweep file (test-time pipelines do not shuffle: PointShuffle is a
    train-only transform in the reference configs): the head turns from az_lo to az_hi and at every azimuth step
    all rings fire, so point i belongs to step i // n_rings, ring i % n_rings."""
    elev = np.deg2rad(np.linspace(elev_lo_deg, elev_hi_deg, n_rings))
    ring = np.arange(n_points) % n_rings
    step = np.arange(n_points) // n_rings
    n_steps = max(int(step[-1]) + 1, 1) if n_points else 1
    az = az_lo + (az_hi - az_lo) * (step + rng.uniform(0.0, 1.0, size=n_points)) / n_steps
    e = elev[ring]
    down = e < -1e-3
    r_ground = np.where(down, sensor_h / np.tan(np.where(down, -e, 1.0)), np.inf)
    r = np.where(down & (r_ground < max_ground_range), r_ground,
                 rng.uniform(up_range[0], up_range[1], size=n_points))
    r = r * rng.normal(1.0, 0.02, size=n_points)
    x = r * np.cos(e) * np.cos(az)
    y = r * np.cos(e) * np.sin(az)
    z = r * np.sin(e)
    return x, y, z


def _clip_open(x, y, z, rng_box, eps=1e-3):
    lo = np.array(rng_box[:3]) + eps
    hi = np.array(rng_box[3:]) - eps
    keep = (x > lo[0]) & (x < hi[0]) & (y > lo[1]) & (y < hi[1]) & (z > lo[2]) & (z < hi[2])
    return keep


def nuscenes_sweep(seed=2000, n_points=30000):
    """(N,5) f32: x, y, z, intensity in [0,255], dt=0.  32-ring 360 degree model."""
    rng = np.random.default_rng(seed)
    n = n_points
    while True:  # one full revolution holding at least n_points in-range returns, thinned to exactly n_points
        x, y, z = _ring_sweep(rng, n, 32, -30.67, 10.67, 1.84, 54.0, (5.0, 50.0), 0.0, 2 * np.pi)
        keep = np.nonzero(_clip_open(x, y, z, NUSC_RANGE))[0]
        if keep.size >= n_points:
            break
        n = int(n * 1.15) + 64
    keep = keep[np.sort(rng.choice(keep.size, n_points, replace=False))]
    inten = rng.uniform(0, 255, size=n_points)
    return np.ascontiguousarray(np.stack([x[keep], y[keep], z[keep], inten, np.zeros(n_points)], 1).astype(np.float32))


def kitti_sweep(seed=1000, n_points=17000):
    """(N,4) f32: x, y, z, reflectance in [0,1].  64-ring front field of view."""
    rng = np.random.default_rng(seed)
    n = n_points
    while True:
        x, y, z = _ring_sweep(rng, n, 64, -24.8, 2.0, 1.73, 70.0, (5.0, 60.0), -np.pi / 4, np.pi / 4)
        keep = np.nonzero(_clip_open(x, y, z, KITTI_RANGE))[0]
        if keep.size >= n_points:
            break
        n = int(n * 1.15) + 64
    keep = keep[np.sort(rng.choice(keep.size, n_points, replace=False))]
    refl = rng.uniform(0, 1, size=n_points)
    return np.ascontiguousarray(np.stack([x[keep], y[keep], z[keep], refl], 1).astype(np.float32))


def waymo_sweep(seed=5000, n_points=180000):
    """(N,5) f32: x, y, z, intensity, elongation.  64-ring model to 75 m."""
    rng = np.random.default_rng(seed)
    n = n_points
    while True:
        x, y, z = _ring_sweep(rng, n, 64, -17.6, 2.4, 1.9, 75.0, (5.0, 70.0), 0.0, 2 * np.pi)
        keep = np.nonzero(_clip_open(x, y, z, WAYMO_RANGE))[0]
        if keep.size >= n_points:
            break
        n = int(n * 1.15) + 64
    keep = keep[np.sort(rng.choice(keep.size, n_points, replace=False))]
    inten, elong = rng.uniform(0, 1, size=n_points), rng.uniform(0, 1, size=n_points)
    return np.ascontiguousarray(np.stack([x[keep], y[keep], z[keep], inten, elong], 1).astype(np.float32))


def camera_rig(n_cam=6, f=1266.0, cx=816.0, cy=491.0, cam_h=1.5):
    """(n_cam,4,4) f32 lidar->image matrices of a pinhole ring (yaws 0, +-55, +-110, 180 deg)."""
    yaws = np.deg2rad([0.0, 55.0, -55.0, 110.0, -110.0, 180.0])[:n_cam]
    K = np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    mats = []
    for yaw in yaws:
        c, s = np.cos(yaw), np.sin(yaw)
        # camera axes in the lidar frame: z_cam = forward, x_cam = right, y_cam = down
        fwd = np.array([c, s, 0.0])
        right = np.array([s, -c, 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd], 0)
        t = -R @ np.array([0.0, 0.0, cam_h])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = t
        mats.append(K @ E)
    return np.stack(mats, 0).astype(np.float32)


def camera_images(seed=3000, n_cam=6, h=928, w=1600):
    """(1,n_cam,3,h,w) f32 ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((1, n_cam, 3, h, w), dtype=np.float32)


def sha256_of(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
