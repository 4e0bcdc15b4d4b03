"""A few EAGER frames of the LiDAR path (hard voxelization -> bitmap rulebooks -> sparse encoder -> densify -> SECOND -> FPN -> decoder with
its RoI gathers) and nothing else: the target of the `rocprofv3 --pmc` passes of tools/measure_traffic.py for the scatter / gather
stages (srf_hv_*, srf_bm_*, srf_densify_k, srf_roi_extract_k).  python tools/prof_lidar_frame.py [frames] [workload]
(workload: nusc_L (default) | waymo_L -- the 180k-point frame of configs/waymo/srfdet_dvoxel_waymo_L.py, dynamic voxelization + scatter)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, randomize_bn  # noqa: E402
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "nusc_L"]
torch.manual_seed(0)
model = workloads.build(wl["cfg"], 200).eval()
randomize_bn(model)
model = model.cuda()
metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
with torch.no_grad():
    for i in range(frames):
        pts = torch.from_numpy(getattr(synthetic, wl.get("sweep", "nuscenes_sweep"))(wl.get("seed", 2000) + i, wl.get("points", 30000))).cuda()
        res = model.simple_test(None, [pts], metas)
torch.cuda.synchronize()
print("done", len(res))
