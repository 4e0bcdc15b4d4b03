"""Per-graph start / end times of the segmented frame (needs tools/micro/frame_segments_experiment.patch applied)."""
import os, sys, torch
sys.path.insert(0, '.')
from srfdet3d_amd import synthetic, workloads, graphs
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = workloads.build("srfdet_voxel_nusc_L", 200).eval().to(dev)
m.enable_hip_graphs()
metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
pts = [torch.from_numpy(synthetic.nuscenes_sweep(i, 30000)).to(dev) for i in range(3)]
with torch.no_grad():
    for i in range(6):
        m.simple_test(None, [pts[i % 3]], metas)
    torch.cuda.synchronize()
    graphs.FrameSegments.debug_times = []
    for i in range(5):
        m.simple_test(None, [pts[i % 3]], metas)
for fr in graphs.FrameSegments.debug_times[-3:]:
    print(" | ".join(f"{k} {a:7.1f}-{b:7.1f}" for k, a, b in fr))
