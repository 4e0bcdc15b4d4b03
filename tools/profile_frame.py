#!/usr/bin/env python3
"""Per-phase wall time of one frame (host clock with syncs; developer tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402


def main():
    torch.manual_seed(0)
    m = workloads.build("srfdet_voxel_nusc_L", 200).eval().cuda()
    m.enable_hip_graphs()
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    frames = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + i)).cuda() for i in range(4)]
    for f in frames:
        with torch.no_grad():
            m.simple_test(None, [f], metas)
    acc = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        t = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t - t0)
        return t

    n = 20
    for i in range(n):
        pts = frames[i % 4]
        with torch.no_grad():
            torch.cuda.synchronize()
            t = time.perf_counter()
            voxels, num, coors = m.voxelize([pts])
            vf = m.pts_voxel_encoder(voxels, num, coors)
            t = lap("voxelize+vfe", t)
            bev = m.pts_middle_encoder(vf, coors, 1)
            t = lap("sparse encoder", t)
            scores, boxes = m._graphed_tail(bev, None, metas)
            t = lap("graph tail (SECOND+FPN+head+decode)", t)
            res = m.bbox_head.get_bboxes(None, None, metas, decoded=(scores, boxes))
            t = lap("nms+filter", t)
            out = [r.to("cpu") if hasattr(r, "to") else r for r in res[0]]
            t = lap("results to host", t)
    for k, v in acc.items():
        print(f"{k:40s} {v / n * 1e3:7.3f} ms")
    print(f"{'sum':40s} {sum(acc.values()) / n * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
