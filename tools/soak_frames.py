"""Soak: several hundred frames of varying size through the graphed path (developer tool): memory must stay flat, the graph
statistics must show replays with the occasional eager fallback + recapture when a sweep exceeds a capacity, and the detections of a
repeated frame must not drift.  python tools/soak_frames.py [nusc_L|nusc_LC] [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "nusc_L"
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(0)
m = workloads.build(bench.WORKLOADS[wl]["cfg"], 200).eval()
bench.randomize_bn(m)
m = m.cuda().enable_hip_graphs(img_overlap=m.use_img, whole_frame=True)
metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
img = None
if m.use_img:
    img = torch.from_numpy(synthetic.camera_images(3000)).cuda()
    metas[0]["lidar2img"] = [x for x in synthetic.camera_rig()]
rng = np.random.default_rng(0)
frames = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + i, int(rng.integers(24000, 40000)))).cuda() for i in range(12)]
first = None
for i in range(n_frames):
    with torch.no_grad():
        out = m.simple_test(img, [frames[i % 12]], [dict(mm) for mm in metas])[0]["pts_bbox"]
    if i % 12 == 0 and i >= 24:   # frame 0 again: same detections as its first graphed run
        sig = (out["scores_3d"].numel(), float(out["scores_3d"].sum()) if out["scores_3d"].numel() else 0.0)
        if first is None:
            first = sig
        assert sig[0] == first[0] and abs(sig[1] - first[1]) <= 1e-3 * max(1.0, abs(first[1])), (i, sig, first)
    if i in (20, n_frames // 3, n_frames - 1):
        torch.cuda.synchronize()
        print(i, "allocated MB", torch.cuda.memory_allocated() // 2**20, "reserved MB", torch.cuda.memory_reserved() // 2**20,
              m._graphed_frame.stats, flush=True)
print("soak ok", wl, n_frames, "frames; repeated frame signature", first)
