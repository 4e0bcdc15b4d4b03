import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from srfdet3d_amd import synthetic, workloads
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
torch.manual_seed(0)
m = workloads.build("srfdet_voxel_nusc_L", 200).eval(); bench.randomize_bn(m); m = m.cuda().enable_hip_graphs()
metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
import numpy as np
rng = np.random.default_rng(0)
frames = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + i, int(rng.integers(24000, 40000)))).cuda() for i in range(12)]
for i in range(300):
    with torch.no_grad():
        m.simple_test(None, [frames[i % 12]], metas)
    if i in (20, 100, 299):
        torch.cuda.synchronize()
        print(i, "allocated MB", torch.cuda.memory_allocated() // 2**20, "reserved MB", torch.cuda.memory_reserved() // 2**20, m._graphed_frame.stats)
