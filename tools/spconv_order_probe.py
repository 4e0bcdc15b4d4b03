"""Kernel-level effect of the mask-sorted row plan (srf_spconv_order_build) on the 32-channel sparse convolutions of a workload:
    python tools/spconv_order_probe.py srfdet_voxel_nusc_L | srfdet_dvoxel_waymo_L
prints plain / with-plan / plan-build microseconds per rulebook and checks that the outputs are equal."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from srfdet3d_amd import ops, synthetic, workloads
dev = torch.device('cuda:0')
torch.manual_seed(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "srfdet_voxel_nusc_L"
m = workloads.build(wl, 200).eval().to(dev)
npts = 180000 if "waymo" in wl else 30000
gen = synthetic.waymo_sweep if "waymo" in wl and hasattr(synthetic, "waymo_sweep") else synthetic.nuscenes_sweep
pts = torch.from_numpy(gen(0, npts)).to(dev)
recs = []
orig = ops.spconv_fwd
def spy(feats, weight, nbr, *a, **k):
    recs.append((feats.shape, tuple(weight.shape), nbr))
    return orig(feats, weight, nbr, *a, **k)
ops.spconv_fwd = spy
with torch.no_grad():
    m.extract_point_features([pts])
ops.spconv_fwd = orig
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
seen = set()
for shape, wshape, nbr in recs:
    K, Cin, Cout = wshape
    if Cout != 32 or (Cin, nbr.data_ptr()) in seen: continue
    seen.add((Cin, nbr.data_ptr()))
    A_in = shape[0]; A_out = nbr.shape[1]
    feats = torch.randn(A_in, Cin, device=dev)
    w = torch.randn(K, Cin, Cout, device=dev) * 0.1
    packed = ops.pack_spconv_weights(w)
    res = torch.randn(A_out, Cout, device=dev)
    t0 = timeit(lambda: ops.spconv_fwd(feats, w, nbr, None, None, res, True, packed=packed))
    plan = ops.spconv_order(nbr)
    t1 = timeit(lambda: ops.spconv_fwd(feats, w, nbr, None, None, res, True, packed=packed, tiles=plan))
    tb = timeit(lambda: ops.spconv_order(nbr))
    a = ops.spconv_fwd(feats, w, nbr, None, None, res, True, packed=packed)
    b = ops.spconv_fwd(feats, w, nbr, None, None, res, True, packed=packed, tiles=plan)
    print(f"{wl} {Cin}->{Cout} A_in {A_in} A_out {A_out}: plain {t0:.1f} us, with plan {t1:.1f} us, plan build {tb:.1f} us, equal {torch.equal(a, b)}")
