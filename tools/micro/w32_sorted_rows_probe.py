"""What a mask-sorted row order would do to srf_spconv_w32_k, before any kernel support: permuted rulebooks fed to the unchanged
kernel (outputs land in permuted rows): current order / centre offset only (fixed cost) / key-sorted / sorted + interleaved."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from srfdet3d_amd import ops, synthetic, workloads
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = workloads.build("srfdet_voxel_nusc_L", 200).eval().to(dev)
pts = torch.from_numpy(synthetic.nuscenes_sweep(0, 30000)).to(dev)
# run the encoder once eagerly to get the level-2 rulebook
from srfdet3d_amd import sparse
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
for shape, wshape, nbr in recs:
    K, Cin, Cout = wshape
    if Cout != 32: continue
    A_in = shape[0]; A_out = nbr.shape[1]
    feats = torch.randn(A_in, Cin, device=dev)
    w = torch.randn(K, Cin, Cout, device=dev) * 0.1
    packed = ops.pack_spconv_weights(w)
    cnt = (nbr >= 0).sum().item()
    t_cur = timeit(lambda: ops.spconv_fwd(feats, w, nbr, packed=packed))
    # only the centre offset
    nb1 = torch.full_like(nbr, -1); nb1[13] = nbr[13]
    t_one = timeit(lambda: ops.spconv_fwd(feats, w, nb1, packed=packed))
    # mask-sorted rows (permute columns of nbr): the MFMA work the sorted form would execute, outputs land in permuted rows
    mask = torch.zeros(A_out, dtype=torch.int64, device=dev)
    for k in range(K): mask |= (nbr[k] >= 0).long() << k
    p = [(mask >> (9 * i)) & 0x1ff for i in range(3)]
    key = (((p[0] != 0).long() | ((p[2] != 0).long() << 1)) << 9) | p[1]
    order = torch.argsort(key, stable=True)
    nbs = nbr[:, order].contiguous()
    t_sorted = timeit(lambda: ops.spconv_fwd(feats, w, nbs, packed=packed))
    # sorted + interleaved groups of 32 (static balance): octiles
    ng = (A_out + 31) // 32; nt = (ng + 7) // 8
    pos = torch.full((nt * 8 * 32,), -1, dtype=torch.long, device=dev)
    g = torch.arange(nt * 8, device=dev); b = g // 8; wv = g % 8
    o = torch.where(wv < 4, wv, 11 - wv)
    src_group = o * nt + b                         # group of the sorted order this (block, wave) takes
    idx = (src_group[:, None] * 32 + torch.arange(32, device=dev)[None]).reshape(-1)
    valid = idx < A_out
    cols = torch.where(valid, order[idx.clamp(max=A_out - 1)], torch.zeros_like(idx))
    nbi = torch.where(valid[None], nbr[:, cols], torch.full((1,), -1, dtype=nbr.dtype, device=dev)).contiguous()
    t_inter = timeit(lambda: ops.spconv_fwd(feats, w, nbi, packed=packed))
    print(f"{Cin}->{Cout} A_in {A_in} A_out {A_out} pairs/row {cnt / A_out:.2f}: current {t_cur:.1f} us, centre only {t_one:.1f}, "
          f"keyA-sorted blocked {t_sorted:.1f}, sorted+interleaved {t_inter:.1f} (rows {nbi.shape[1]})")
