import sys, time, cProfile, pstats, torch
sys.path.insert(0, "/root/repo")
import bench
from srfdet3d_amd import synthetic, workloads
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
torch.manual_seed(0)
m = workloads.build(bench.WORKLOADS["nusc_LC"]["cfg"], 200).eval(); bench.randomize_bn(m); m = m.cuda()
m.enable_hip_graphs(whole_frame=True)
img = torch.from_numpy(synthetic.camera_images(3000)).cuda()
frames = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + i, 30000)).cuda() for i in range(4)]
def metas(): return [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[x for x in synthetic.camera_rig()])]
mt = metas()
with torch.no_grad():
    for i in range(6): m.simple_test(img, [frames[i % 4]], mt)
    torch.cuda.synchronize()
    # wall-clock stamps inside one frame
    import srfdet3d_amd.graphs as G
    orig_replay = torch.cuda.CUDAGraph.replay
    stamps = []
    def replay(self):
        t0 = time.perf_counter(); orig_replay(self); stamps.append(("replay", t0, time.perf_counter()))
    torch.cuda.CUDAGraph.replay = replay
    for i in range(5):
        stamps.clear()
        t0 = time.perf_counter()
        m.simple_test(img, [frames[i % 4]], mt)
        t1 = time.perf_counter()
        print("frame %.3f ms:" % ((t1 - t0) * 1e3), " ".join("%s@%.3f+%.3f" % (n, (a - t0) * 1e3, (b - a) * 1e3) for n, a, b in stamps))
    torch.cuda.CUDAGraph.replay = orig_replay
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for i in range(30): m.simple_test(img, [frames[i % 4]], mt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pr.disable()
print("ms/frame", (t1 - t0) / 30 * 1e3)
ps = pstats.Stats(pr); ps.sort_stats("cumulative").print_stats(30)
