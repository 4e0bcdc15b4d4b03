"""Size-independent properties at BASELINE.json's largest shapes (Waymo: 180k points, 1536 x 1536 x 41 grid), where a
direct oracle comparison of every intermediate would be slow: invariants that must hold whatever the input."""
import numpy as np
import pytest
import torch

from srfdet3d_amd import ops, synthetic as S, workloads

pytestmark = pytest.mark.gpu
WAYMO = dict(voxel_size=[0.1, 0.1, 0.15], pc_range=list(S.WAYMO_RANGE))


def test_hard_voxelize_conservation_full_waymo(dev):
    pts = S.waymo_sweep(5000)
    p = torch.from_numpy(pts).to(dev)
    voxels, coors, num, mean = ops.hard_voxelize(p, WAYMO["voxel_size"], WAYMO["pc_range"], 10, 400000, mean_features=5)
    c = coors.cpu().numpy().astype(np.int64)
    key = (c[:, 0] * 1536 + c[:, 1]) * 1536 + c[:, 2]
    assert len(np.unique(key)) == len(key), "a voxel appears once"
    dyn = ops.dynamic_voxelize(p, WAYMO["voxel_size"], WAYMO["pc_range"]).cpu().numpy().astype(np.int64)
    dkey = (dyn[:, 0] * 1536 + dyn[:, 1]) * 1536 + dyn[:, 2]
    # the voxel set equals the set of cells the points fall in, and counts are min(points in cell, max_points)
    uniq, cnt = np.unique(dkey[dyn[:, 0] >= 0], return_counts=True)
    order = np.argsort(key)
    np.testing.assert_array_equal(key[order], uniq)
    np.testing.assert_array_equal(num.cpu().numpy()[order], np.minimum(cnt, 10))
    # first-seen order: the first point of voxel m comes before the first point of voxel m+1
    first = {}
    for i, k in enumerate(dkey):
        if k >= 0 and k not in first:
            first[k] = i
    f = np.array([first[k] for k in key])
    assert (np.diff(f) > 0).all()
    # every stored point lies in its voxel and slot 0 holds the voxel's first point
    v = voxels.cpu().numpy()
    np.testing.assert_array_equal(v[:, 0, :], pts[f])
    s = v.sum(1) / num.cpu().numpy()[:, None]
    np.testing.assert_allclose(mean.cpu().numpy(), s, rtol=1e-5, atol=1e-4)


def test_encoder_is_invariant_to_row_order_full_waymo(dev):
    """the dense BEV map must not depend on the order of the active sites (spatial ordering on/off, random shuffle)."""
    torch.manual_seed(0)
    model = workloads.build("srfdet_dvoxel_waymo_L", 16).eval().to(dev)
    pts = torch.from_numpy(S.waymo_sweep(5000)).to(dev)
    with torch.no_grad():
        p, coors = model.voxelize([pts])
        vf, vc = model.pts_voxel_encoder(p, coors)
        enc = model.pts_middle_encoder
        a = enc(vf, vc, 1)
        enc.spatial_sort = False
        b = enc(vf, vc, 1)
        perm = torch.randperm(vc.shape[0], device=dev)
        c = enc(vf[perm].contiguous(), vc[perm].contiguous(), 1)
        enc.spatial_sort = True
    assert a.shape == (1, 256, 192, 192)
    assert torch.equal(a, b) and torch.equal(a, c)
    assert int((a != 0).sum()) > 0


def test_spconv_is_linear_without_activation(dev):
    g = torch.Generator().manual_seed(0)
    pts = torch.from_numpy(S.waymo_sweep(5001, 60000)).to(dev)
    _, c, _, _ = ops.hard_voxelize(pts, WAYMO["voxel_size"], WAYMO["pc_range"], 5, 400000)
    idx = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32, device=dev), c], 1).contiguous()
    shape = [41, 1536, 1536]
    nbr, _ = ops.rulebook_subm(idx, shape, [3, 3, 3], ops.coord_table_build(idx, shape, 1))
    A = idx.shape[0]
    x, y = torch.randn(A, 32, generator=g).to(dev), torch.randn(A, 32, generator=g).to(dev)
    W = (torch.randn(27, 32, 64, generator=g) * 0.1).to(dev)
    pk = ops.pack_spconv_weights(W)
    fx, fy, fxy = (ops.spconv_fwd(t, W, nbr, packed=pk) for t in (x, y, 2.0 * x - y))
    torch.testing.assert_close(fxy, 2.0 * fx - fy, rtol=1e-4, atol=1e-4)
    # the centre tap alone is a plain matrix product
    W0 = torch.zeros_like(W)
    W0[13] = W[13]
    torch.testing.assert_close(ops.spconv_fwd(x, W0, nbr), x @ W[13], rtol=1e-4, atol=1e-4)
