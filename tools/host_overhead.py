import sys, time, torch
sys.path.insert(0, '/root/repo')
from srfdet3d_amd import ops, synthetic, workloads
from srfdet3d_amd.sparse import SparseConvTensor
torch.manual_seed(0)
m = workloads.build("srfdet_voxel_nusc_L", 200).eval().cuda()
enc = m.pts_middle_encoder
pts = torch.from_numpy(synthetic.nuscenes_sweep(2000, 3000)).cuda()
with torch.no_grad():
    voxels, num, coors = m.voxelize([pts])
    vf = m.pts_voxel_encoder(voxels, num, coors)
    for _ in range(3):
        enc(vf, coors, 1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        enc(vf, coors, 1)
    t1 = time.perf_counter() - t
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t
    print(f"encoder host time per frame (tiny input, {coors.shape[0]} voxels): {t1/20*1e3:.3f} ms; with final sync {t2/20*1e3:.3f} ms")
    x = SparseConvTensor.sorted_by_bitmap(vf, coors.int(), enc.sparse_shape, 1)
    x = enc.conv_input(x)
    blk = enc.encoder_layers.encoder_layer1[0]
    for _ in range(3):
        blk(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        blk(x)
    t1 = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"SparseBasicBlock (2 fused convs) host time: {t1/200*1e6:.1f} us")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        enc(vf, coors, 1)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
