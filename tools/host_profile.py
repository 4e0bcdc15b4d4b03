import sys, time, copy, cProfile, pstats, torch
sys.path.insert(0, "/root/repo")
import bench
from srfdet3d_amd import synthetic, workloads
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = workloads.build(bench.WORKLOADS["nusc_L"]["cfg"], 200).eval(); bench.randomize_bn(m); m = m.to(dev)
m.enable_hip_graphs(whole_frame=True)
frames = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + i, 30000)).to(dev) for i in range(4)]
metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
with torch.no_grad():
    for i in range(10): m.simple_test(None, [frames[i % 4]], metas)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for i in range(200): m.simple_test(None, [frames[i % 4]], metas)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pr.disable()
print("ms/frame", (t1 - t0) / 200 * 1e3)
ps = pstats.Stats(pr); ps.sort_stats("cumulative").print_stats(28)
