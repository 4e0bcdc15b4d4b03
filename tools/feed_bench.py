#!/usr/bin/env python3
"""PCIe-inclusive rate of a workload (developer tool; `bench.py` keeps the inputs resident in HBM as its contract says).

Per frame: the raw sweep (float32) and the decoded camera views (uint8, (V, 900, 1600, 3)) start in pinned host memory,
are uploaded on a copy stream one frame ahead of the compute (srfdet3d_amd.plugin.pipelines.FrameFeeder), go through the
device-side test pipeline (PointsRangeFilter; NormalizeMultiviewImage + PadMultiViewImage) and then through the model.
Also prints the two preparation kernels against the HBM rate.
usage: python tools/feed_bench.py [--workload nusc_LC|nusc_L] [--steps 30] [--warmup 8]"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops, synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402
from srfdet3d_amd.plugin.pipelines import Compose, FrameFeeder  # noqa: E402

IMG_NORM = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nusc_LC", choices=["nusc_LC", "nusc_L"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = dict(nusc_LC="srfdet_voxel_nusc_LC", nusc_L="srfdet_voxel_nusc_L")[a.workload]
    torch.manual_seed(0)
    model = workloads.build(cfg, 200).eval()
    from bench import randomize_bn
    randomize_bn(model)
    model = model.to(dev)
    model.enable_hip_graphs(img_overlap=model.use_img, whole_frame=True)   # the configuration bench.py runs by default
    pc_range = list(model.pts_voxel_layer.point_cloud_range) if hasattr(model, "pts_voxel_layer") else list(synthetic.NUSC_RANGE)
    V, H, W = 6, 900, 1600
    rng = np.random.default_rng(0)
    host_pts = [synthetic.nuscenes_sweep(2000 + i, 30000) for i in range(4)]
    host_img = [rng.integers(0, 256, (V, H, W, 3), dtype=np.uint8) for _ in range(2)] if model.use_img else None
    pts_pipe = Compose([dict(type="PointsRangeFilter", point_cloud_range=pc_range)])
    img_pipe = Compose([dict(type="NormalizeMultiviewImage", **IMG_NORM), dict(type="PadMultiViewImage", size_divisor=32)])
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    if model.use_img:
        metas[0]["lidar2img"] = [m for m in synthetic.camera_rig()]
    feeder = FrameFeeder(40000, 5, V, H, W, dev)

    def frame(i, slot_holder):
        pts, img_u8, slot = feeder.get()
        nxt = i + 1
        feeder.put(host_pts[nxt % 4], host_img[nxt % 2] if host_img else None)   # upload of the next frame overlaps this one
        res = pts_pipe(dict(points=pts))
        img = None
        if model.use_img:
            img = img_pipe(dict(img=img_u8))["img"].unsqueeze(0)  # the collate's batch dimension: (1, V, 3, Hp, Wp)
        FrameFeeder.release(slot)
        with torch.no_grad():
            return model.simple_test(img, [res["points"]], metas)

    feeder.put(host_pts[0], host_img[0] if host_img else None)
    for i in range(a.warmup):
        frame(i, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        frame(a.warmup + i, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # the two preparation kernels on their own
    d_img = torch.from_numpy(host_img[0]).to(dev) if host_img else torch.zeros((V, H, W, 3), dtype=torch.uint8, device=dev)
    d_pts = torch.from_numpy(host_pts[0]).to(dev)
    us_img = timed(lambda: ops.image_prepare(d_img, IMG_NORM["mean"], IMG_NORM["std"], False, 32))
    img_bytes = V * H * W * 3 + V * 3 * 928 * 1600 * 4
    us_pts = timed(lambda: ops.points_filter(d_pts, pc_range, static=True))
    big = torch.from_numpy(synthetic.waymo_sweep(5000, 180000)).to(dev)
    us_big = timed(lambda: ops.points_filter(big, list(synthetic.WAYMO_RANGE), static=True))
    gf = getattr(model, "_graphed_frame", None)
    print(json.dumps({
        "graph_stats": dict(gf.stats) if gf is not None else None,
        "workload": cfg, "fed_from_host": True, "frames_per_s": round(a.steps / dt, 3), "ms_per_frame": round(dt / a.steps * 1e3, 3),
        "h2d_bytes_per_frame": int(host_pts[0].nbytes + (host_img[0].nbytes if host_img else 0)),
        "image_prepare": {"us": round(us_img, 1), "bytes": img_bytes, "GB_per_s": round(img_bytes / us_img / 1e3, 1),
                          "frac_of_8TBps": round(img_bytes / us_img / 1e3 / 8000, 3)},
        "points_filter_30k": {"us": round(us_pts, 1), "launches": 3},
        "points_filter_180k": {"us": round(us_big, 1), "bytes": int(big.numel() * 4 * 2), "GB_per_s": round(big.numel() * 8 / us_big / 1e3, 1)},
    }))


if __name__ == "__main__":
    main()
