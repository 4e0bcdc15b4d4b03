"""The C oracle (oracle/srf_oracle.c) against brute-force definitions of the third-party operators
 on small inputs, and against torch CPU where torch has the same op.
These operators are "parity unpinned" w.r.t. the reference (it ships no fixtures for them)."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import oracle as O

VS, RANGE = [0.5, 0.5, 1.0], [0.0, -4.0, -2.0, 8.0, 4.0, 2.0]  # grid 16 x 16 x 4


def _brute_hard(points, max_points, max_voxels):
    grid = O.grid_size(VS, RANGE)
    lut, voxels, coors, num = {}, [], [], []
    for p in points:
        c = np.floor((p[:3] - np.float32(RANGE[:3])) / np.float32(VS)).astype(np.int64)
        if (c < 0).any() or (c >= grid).any():
            continue
        key = (c[2], c[1], c[0])
        if key not in lut:
            if len(voxels) >= max_voxels:
                continue
            lut[key] = len(voxels)
            voxels.append(np.zeros((max_points, points.shape[1]), np.float32))
            coors.append(key)
            num.append(0)
        i = lut[key]
        if num[i] < max_points:
            voxels[i][num[i]] = p
            num[i] += 1
    return (np.array(voxels, np.float32).reshape(-1, max_points, points.shape[1]),
            np.array(coors, np.int32).reshape(-1, 3), np.array(num, np.int32))


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 400), st.integers(1, 5), st.integers(1, 60))
def test_hard_voxelize_definition(seed, n, max_points, max_voxels):
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-1, -5, -3, 0], [9, 5, 3, 1], (n, 4)).astype(np.float32)
    pts[::7, :3] = np.float32(RANGE[:3]) + np.float32(VS) * rng.integers(0, 4, (len(pts[::7]), 3))  # on voxel faces
    v, c, k = O.hard_voxelize(pts, VS, RANGE, max_points, max_voxels)
    bv, bc, bk = _brute_hard(pts, max_points, max_voxels)
    np.testing.assert_array_equal(c, bc)
    np.testing.assert_array_equal(k, bk)
    np.testing.assert_array_equal(v, bv)
    if len(k):
        np.testing.assert_allclose(O.vfe_mean(v, k), v.sum(1) / k[:, None], rtol=1e-6, atol=1e-6)
    dyn = O.dynamic_voxelize(pts, VS, RANGE)
    inside = dyn[:, 0] >= 0
    assert inside.sum() >= k.sum()
    assert ((dyn[~inside] == -1).all())


def test_dynamic_scatter_definition():
    rng = np.random.default_rng(0)
    n = 500
    coors = np.concatenate([rng.integers(0, 2, (n, 1)), rng.integers(0, 4, (n, 1)), rng.integers(0, 6, (n, 2))], 1).astype(np.int32)
    coors[::9] = -1
    feats = rng.standard_normal((n, 3)).astype(np.float32)
    for mode in ("mean", "max"):
        f, c, p2v = O.dynamic_scatter(feats, coors, [4, 6, 6], mode)
        valid = coors[:, 0] >= 0
        uniq, inv = np.unique(coors[valid], axis=0, return_inverse=True)  # lexicographic, like torch.unique(dim=0)
        np.testing.assert_array_equal(c, uniq)
        np.testing.assert_array_equal(p2v[valid], inv.ravel())
        assert (p2v[~valid] == -1).all()
        for m in range(len(uniq)):
            rows = feats[valid][inv.ravel() == m]
            want = rows.mean(0) if mode == "mean" else rows.max(0)
            np.testing.assert_allclose(f[m], want, rtol=1e-5, atol=1e-6)


def _active(rng, shape, batch, p):
    occ = rng.random((batch, *shape)) < p
    idx = np.argwhere(occ).astype(np.int32)
    return idx[rng.permutation(len(idx))]


def test_rulebooks_definition():
    rng = np.random.default_rng(1)
    shape = [7, 10, 9]
    idx = _active(rng, shape, 2, 0.2)
    where = {tuple(c): i for i, c in enumerate(idx)}
    nbr, cnt = O.rulebook_subm(idx, shape, [3, 3, 3])
    for o, c in enumerate(idx):
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    k = (kz * 3 + ky) * 3 + kx
                    want = where.get((c[0], c[1] + kz - 1, c[2] + ky - 1, c[3] + kx - 1), -1)
                    assert nbr[k, o] == want
    assert cnt.sum() == (nbr >= 0).sum()
    for ks, stv, pd in (([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]), ([3, 1, 1], [2, 1, 1], [0, 0, 0])):
        oi, nbr, cnt, osh = O.rulebook_strided(idx, shape, ks, stv, pd)
        assert osh == [(shape[d] + 2 * pd[d] - ks[d]) // stv[d] + 1 for d in range(3)]
        # active outputs = every q reached by some input p = q*s - pad + k (Appendix B.4), each exactly once
        want = set()
        for c in idx:
            for kz in range(ks[0]):
                for ky in range(ks[1]):
                    for kx in range(ks[2]):
                        t = [c[1] + pd[0] - kz, c[2] + pd[1] - ky, c[3] + pd[2] - kx]
                        if all(t[d] >= 0 and t[d] % stv[d] == 0 and t[d] // stv[d] < osh[d] for d in range(3)):
                            want.add((c[0], t[0] // stv[0], t[1] // stv[1], t[2] // stv[2]))
        assert set(map(tuple, oi)) == want and len(oi) == len(want)
        for k in range(nbr.shape[0]):
            kz, ky, kx = k // (ks[1] * ks[2]), (k // ks[2]) % ks[1], k % ks[2]
            for o, q in enumerate(oi):
                p = (q[0], q[1] * stv[0] - pd[0] + kz, q[2] * stv[1] - pd[1] + ky, q[3] * stv[2] - pd[2] + kx)
                assert nbr[k, o] == where.get(p, -1)


@pytest.mark.parametrize("ks,stv,pd", [([3, 3, 3], [1, 1, 1], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                        ([3, 1, 1], [2, 1, 1], [0, 0, 0])])
def test_spconv_vs_torch_conv3d(ks, stv, pd):
    rng = np.random.default_rng(2)
    shape = [9, 12, 10]
    idx = _active(rng, shape, 2, 0.25)
    cin, cout = 6, 8
    feats = rng.standard_normal((len(idx), cin)).astype(np.float32)
    K = int(np.prod(ks))
    W = (rng.standard_normal((K, cin, cout)) * 0.2).astype(np.float32)
    subm = stv == [1, 1, 1]
    if subm:
        nbr, _ = O.rulebook_subm(idx, shape, ks)
        oi, osh = idx, shape
    else:
        oi, nbr, _, osh = O.rulebook_strided(idx, shape, ks, stv, pd)
    got = O.spconv_fwd(feats, W, nbr)
    dense = np.zeros((2, cin, *shape), np.float32)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feats
    w5 = torch.from_numpy(W.reshape(*ks, cin, cout)).permute(4, 3, 0, 1, 2).contiguous()
    full = torch.nn.functional.conv3d(torch.from_numpy(dense).double(), w5.double(), stride=stv, padding=pd).numpy()
    assert list(full.shape[2:]) == osh
    np.testing.assert_allclose(got, full[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]], rtol=1e-5, atol=1e-5)
    if not subm:  # a regular sparse conv is active wherever the dense result can be non-zero
        mask = np.zeros((2, *osh), bool)
        mask[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]] = True
        assert np.abs(full).sum(1)[~mask].max(initial=0) == 0
    # fused epilogue = BatchNorm1d(eval) -> (+residual) -> ReLU as torch computes it
    bn = torch.nn.BatchNorm1d(cout, eps=1e-3).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout) + 0.5)
        bn.bias.copy_(torch.randn(cout))
        bn.running_mean.copy_(torch.randn(cout) * 0.1)
        bn.running_var.copy_(torch.rand(cout) + 0.5)
        want = torch.relu(bn(torch.from_numpy(got))).numpy()
    a, b = O.bn_fold(bn.weight.detach().numpy(), bn.bias.detach().numpy(), bn.running_mean.numpy(), bn.running_var.numpy(), bn.eps)
    np.testing.assert_allclose(O.spconv_fwd(feats, W, nbr, a, b, None, True), want, rtol=1e-5, atol=1e-6)
    dz = O.densify(got, oi, 2, osh)
    assert dz.shape == (2, cout, *osh) and np.count_nonzero(dz) <= got.size
    np.testing.assert_array_equal(dz[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]], got)


def _roi_align_def(feat, roi, scale, P=7, sr=2):
    """plain-Python definition of mmcv RoIAlign(avg, aligned=True) for one RoI (Appendix B.5), float64."""
    N, C, H, W = feat.shape
    n = int(roi[0])
    x1, y1, x2, y2 = [float(np.float32(v) * np.float32(scale)) - 0.5 for v in roi[1:]]
    bw, bh = (x2 - x1) / P, (y2 - y1) / P
    out = np.zeros((C, P, P))
    for ph in range(P):
        for pw in range(P):
            acc = np.zeros(C)
            for iy in range(sr):
                for ix in range(sr):
                    y = y1 + ph * bh + (iy + 0.5) * bh / sr
                    x = x1 + pw * bw + (ix + 0.5) * bw / sr
                    if y < -1 or y > H or x < -1 or x > W:
                        continue
                    y, x = max(y, 0.0), max(x, 0.0)
                    yl, xl = int(y), int(x)
                    if yl >= H - 1:
                        yl = yh = H - 1
                        y = float(yl)
                    else:
                        yh = yl + 1
                    if xl >= W - 1:
                        xl = xh = W - 1
                        x = float(xl)
                    else:
                        xh = xl + 1
                    ly, lx = y - yl, x - xl
                    acc += ((1 - ly) * (1 - lx) * feat[n, :, yl, xl] + (1 - ly) * lx * feat[n, :, yl, xh] +
                            ly * (1 - lx) * feat[n, :, yh, xl] + ly * lx * feat[n, :, yh, xh])
            out[:, ph, pw] = acc / (sr * sr)
    return out


def test_roi_align_and_level_mapping_definition():
    rng = np.random.default_rng(3)
    feats = [rng.standard_normal((2, 5, s, s + 3)).astype(np.float32) for s in (40, 20, 10, 5)]
    rois = np.array([[0, 10.2, 20.7, 60.1, 90.3], [1, -30, -20, 40, 35], [0, 100, 100, 700, 650], [1, 5, 5, 5, 5],
                     [0, 300, 300, 301, 302], [1, 0, 0, 330, 330], [0, 900, 900, 950, 980]], np.float32)
    out, lvl = O.roi_extract(feats, rois, [8, 16, 32, 64])
    scale = np.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    want_lvl = np.clip(np.floor(np.log2(scale / 56 + 1e-6)), 0, 3).astype(int)
    np.testing.assert_array_equal(lvl, want_lvl)
    assert len(set(lvl)) >= 3
    for r in range(len(rois)):
        want = _roi_align_def(feats[lvl[r]].astype(np.float64), rois[r], 1.0 / [8, 16, 32, 64][lvl[r]])
        np.testing.assert_allclose(out[r], want, rtol=1e-4, atol=1e-5)
    single = O.roi_align(feats[0], rois, 1 / 8.0)
    np.testing.assert_allclose(single[0], _roi_align_def(feats[0].astype(np.float64), rois[0], 1 / 8.0), rtol=1e-4, atol=1e-5)
