"""Replay time of the LC camera graph alone (VoVNet-99 -> FPN -> img_convs, no LiDAR half beside it) against the whole
frame: how much the BEV half costs the camera branch it runs beside (developer tool).  python tools/cam_graph_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, randomize_bn  # noqa: E402
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402

torch.manual_seed(0)
model = workloads.build(WORKLOADS["nusc_LC"]["cfg"], 200).eval()
randomize_bn(model)
model = model.cuda().enable_hip_graphs(whole_frame=True)
img = torch.from_numpy(synthetic.camera_images(3000)).cuda()
pts = torch.from_numpy(synthetic.nuscenes_sweep(2000, 30000)).cuda()
metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in synthetic.camera_rig()])]
with torch.no_grad():
    for _ in range(4):
        model.simple_test(img, [pts], [dict(m) for m in metas])
    torch.cuda.synchronize()
    for name, fn in (("camera graph alone", lambda: model._graphed_img(img, [dict(m) for m in metas])),
                     ("whole frame", lambda: model.simple_test(img, [pts], [dict(m) for m in metas]))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
            torch.cuda.synchronize()
        print(f"{name:20s} {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms")
