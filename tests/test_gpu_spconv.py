"""GPU parity: K4 rulebooks (bit-exact, including row order), K5 sparse conv (exact fma chain), K6 densify."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from srfdet3d_amd import ops, synthetic as S

pytestmark = pytest.mark.gpu
VS, RANGE = [0.075, 0.075, 0.2], list(S.NUSC_RANGE)
SHAPE1 = [41, 1472, 1472]


def _level1(seed=2000, n=30000, batch=1):
    idx = []
    for b in range(batch):
        _, c, _ = O.hard_voxelize(S.nuscenes_sweep(seed + b, n), VS, RANGE, 10, 160000)
        idx.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return np.concatenate(idx, 0).astype(np.int32)


def _canon(nbr, in_idx, out_idx):
    """canonical rulebook of SURVEY.md 8(a): per k the sorted set of (in coord, out coord) pairs."""
    out = []
    for k in range(nbr.shape[0]):
        o = np.nonzero(nbr[k] >= 0)[0]
        pairs = np.concatenate([in_idx[nbr[k][o]], out_idx[o]], 1)
        out.append(pairs[np.lexsort(pairs.T[::-1])])
    return out


def test_subm_rulebook_bit_exact(dev):
    idx = _level1(batch=2)
    nbr, cnt = O.rulebook_subm(idx, SHAPE1, [3, 3, 3])
    t = torch.from_numpy(idx).to(dev)
    table = ops.coord_table_build(t, SHAPE1, 2)
    gn, gc = ops.rulebook_subm(t, SHAPE1, [3, 3, 3], table)
    np.testing.assert_array_equal(gn.cpu().numpy(), nbr)
    np.testing.assert_array_equal(gc.cpu().numpy(), cnt)
    assert (nbr[13] == np.arange(len(idx))).all()  # centre tap is the identity


def test_strided_rulebook_chain_bit_exact(dev):
    """the four strided convs of the nuScenes encoder, each fed by the previous level (Appendix A)."""
    idx = _level1()
    shape = SHAPE1
    specs = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
             ([3, 1, 1], [2, 1, 1], [0, 0, 0])]
    expect_shapes = [[21, 736, 736], [11, 368, 368], [5, 184, 184], [2, 184, 184]]
    for (ks, st, pd), es in zip(specs, expect_shapes):
        oi, nbr, cnt, osh = O.rulebook_strided(idx, shape, ks, st, pd)
        gi, gn, gc, table, gsh = ops.rulebook_strided(torch.from_numpy(idx).to(dev), shape, 1, ks, st, pd)
        assert osh == es and gsh == es
        np.testing.assert_array_equal(gi.cpu().numpy(), oi)
        np.testing.assert_array_equal(gn.cpu().numpy(), nbr)
        np.testing.assert_array_equal(gc.cpu().numpy(), cnt)
        # canonical form agrees too (order-free statement of the same rulebook)
        for a, b in zip(_canon(gn.cpu().numpy(), idx, gi.cpu().numpy()), _canon(nbr, idx, oi)):
            np.testing.assert_array_equal(a, b)
        # the table built for the output level serves SubM lookups on it
        sn, sc = O.rulebook_subm(oi, osh, [3, 3, 3])
        gsn, gsc = ops.rulebook_subm(gi, osh, [3, 3, 3], table)
        np.testing.assert_array_equal(gsn.cpu().numpy(), sn)
        idx, shape = oi, osh


def test_rulebook_empty_and_single(dev):
    e = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    table = ops.coord_table_build(e, SHAPE1, 1)
    gn, gc = ops.rulebook_subm(e, SHAPE1, [3, 3, 3], table)
    assert gn.shape == (27, 0) and int(gc.sum()) == 0
    gi, gn, gc, _, _ = ops.rulebook_strided(e, SHAPE1, 1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert gi.shape[0] == 0
    one = np.array([[0, 40, 1471, 1471]], np.int32)  # corner voxel: most offsets fall outside the grid
    oi, nbr, cnt, _ = O.rulebook_strided(one, SHAPE1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    gi, gn, gc, _, _ = ops.rulebook_strided(torch.from_numpy(one).to(dev), SHAPE1, 1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    np.testing.assert_array_equal(gi.cpu().numpy(), oi)
    np.testing.assert_array_equal(gn.cpu().numpy(), nbr)


@pytest.mark.parametrize("cin,cout,K", [(5, 16, 27), (16, 16, 27), (16, 32, 27), (32, 32, 27), (32, 64, 27),
                                         (64, 64, 27), (64, 128, 27), (128, 128, 27), (128, 128, 3), (4, 16, 27)])
def test_spconv_fwd_exact(dev, cin, cout, K):
    rng = np.random.default_rng(cin * 1000 + cout)
    idx = _level1(n=6000)
    if K == 27:
        nbr, _ = O.rulebook_subm(idx, SHAPE1, [3, 3, 3])
        a_in = len(idx)
    else:
        oi, nbr, _, _ = O.rulebook_strided(idx, SHAPE1, [3, 1, 1], [2, 1, 1], [0, 0, 0])
        a_in = len(idx)
    feats = rng.standard_normal((a_in, cin)).astype(np.float32)
    W = (rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 3)).astype(np.float32)
    alpha = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    beta = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((nbr.shape[1], cout)).astype(np.float32) if K == 27 else None
    for use_bn, use_res, relu in ((False, False, False), (True, False, True), (True, True, True)):
        if use_res and res is None:
            continue
        ref = O.spconv_fwd(feats, W, nbr, alpha if use_bn else None, beta if use_bn else None,
                           res if use_res else None, relu)
        tt = lambda x: torch.from_numpy(x).to(dev)
        got = ops.spconv_fwd(tt(feats), tt(W), tt(nbr), tt(alpha) if use_bn else None, tt(beta) if use_bn else None,
                             tt(res) if use_res else None, relu).cpu().numpy()
        # exact: both sides are the same f32 fma chain (k ascending, c ascending); == treats +0/-0 alike
        assert np.array_equal(got, ref), f"max abs diff {np.abs(got - ref).max()}"
        if cout >= 32 and cin % 4 == 0:  # packed-weight kernel: same chain, different operand staging
            got = ops.spconv_fwd(tt(feats), tt(W), tt(nbr), tt(alpha) if use_bn else None, tt(beta) if use_bn else None,
                                 tt(res) if use_res else None, relu, packed=ops.pack_spconv_weights(tt(W))).cpu().numpy()
            assert np.array_equal(got, ref), f"packed: max abs diff {np.abs(got - ref).max()}"
            if ops.spconv_tiles_wanted(cin, cout):  # the same kernel walking work-balanced row ranges / a mask-sorted row order
                tiles = ops.spconv_order(tt(nbr)) if ops.spconv_order_wanted(cin, cout) else ops.spconv_tiles(tt(nbr))
                got = ops.spconv_fwd(tt(feats), tt(W), tt(nbr), tt(alpha) if use_bn else None, tt(beta) if use_bn else None,
                                     tt(res) if use_res else None, relu, packed=ops.pack_spconv_weights(tt(W)),
                                     tiles=tiles).cpu().numpy()
                assert np.array_equal(got, ref), f"packed+tiles: max abs diff {np.abs(got - ref).max()}"


@pytest.mark.parametrize("n,live", [(6000, None), (6000, 4100), (40, None), (3000, 0)])
def test_spconv_tiles_cut_rows_by_pair_count(dev, n, live):
    """tiles[t] = first row whose exclusive cost prefix reaches ceil(t * P / T), cost of a row = its pairs +
    srf_spconv_tiles_row_cost() (include/srfdet3d.h): checked against that definition in numpy; with a device row count (static-shape levels) only the live rows
    are cut."""
    idx = _level1(n=n)
    nbr, _ = O.rulebook_subm(idx, SHAPE1, [3, 3, 3])
    A = nbr.shape[1]
    t_nbr = torch.from_numpy(nbr).to(dev)
    rows_dev = None if live is None else torch.tensor([live], dtype=torch.int32, device=dev)
    tiles = ops.spconv_tiles(t_nbr, rows_dev).cpu().numpy()
    a_live = A if live is None else min(live, A)
    T = len(tiles) - 1
    from srfdet3d_amd import _lib
    row_cost = _lib.lib().srf_spconv_tiles_row_cost()
    pairs = (nbr[:, :a_live] >= 0).sum(0).astype(np.int64) + row_cost
    prefix = np.concatenate([[0], np.cumsum(pairs)])  # prefix[r] = cost of rows < r
    P = int(prefix[-1])
    want = np.empty(T + 1, np.int64)
    for t in range(T):
        want[t] = np.searchsorted(prefix[:a_live], -(-P * t // T), side="left") if a_live else 0
    want[T] = a_live
    np.testing.assert_array_equal(tiles, want)
    assert tiles[0] == 0 and np.all(np.diff(tiles) >= 0)
    if a_live >= 2000:  # every range carries P/T cost to within one row's worth
        per = np.diff(prefix[tiles])
        assert per.max() - per.min() <= 2 * (27 + row_cost)


@pytest.mark.parametrize("C", [128, 64])
def test_spconv_tiles_with_padded_rows(dev, C):
    """static-shape level: capacity-sized rulebook whose tail rows are padding (-1), device row count -> same rows out."""
    rng = np.random.default_rng(5)
    idx = _level1(n=5000)
    nbr, _ = O.rulebook_subm(idx, SHAPE1, [3, 3, 3])
    A = nbr.shape[1]
    cap = A + 700
    nbr_pad = np.full((27, cap), -1, np.int32)
    nbr_pad[:, :A] = nbr
    feats = rng.standard_normal((A, C)).astype(np.float32)
    W = (rng.standard_normal((27, C, C)) / 20).astype(np.float32)
    ref = O.spconv_fwd(feats, W, nbr)
    tt = lambda x: torch.from_numpy(x).to(dev)
    rows_dev = torch.tensor([A], dtype=torch.int32, device=dev)
    t_nbr = tt(nbr_pad)
    tiles = ops.spconv_tiles(t_nbr, rows_dev)
    got = ops.spconv_fwd(tt(feats), tt(W), t_nbr, packed=ops.pack_spconv_weights(tt(W)), rows_dev=rows_dev, tiles=tiles)
    assert np.array_equal(got[:A].cpu().numpy(), ref)


@pytest.mark.parametrize("cin", [5, 16])
def test_spconv_c16_lds_weight_form_is_bit_identical(dev, cin):
    """srf_spconv_c16l_k (weights in LDS, five waves per SIMD; taken from 60k rows up) against the oracle and the register-weight form on
    a level large enough to select it (70k rows: three sweeps stacked as a batch of three)"""
    rng = np.random.default_rng(cin)
    idx = _level1(n=30000, batch=3)
    nbr, _ = O.rulebook_subm(idx, SHAPE1, [3, 3, 3])
    assert nbr.shape[1] >= 60000
    feats = rng.standard_normal((len(idx), cin)).astype(np.float32)
    W = (rng.standard_normal((27, cin, 16)) / np.sqrt(cin * 3)).astype(np.float32)
    res = rng.standard_normal((nbr.shape[1], 16)).astype(np.float32)
    alpha = rng.uniform(0.5, 1.5, 16).astype(np.float32)
    beta = rng.standard_normal(16).astype(np.float32)
    ref = O.spconv_fwd(feats, W, nbr, alpha, beta, res, True)
    tt = lambda x: torch.from_numpy(x).to(dev)
    got = ops.spconv_fwd(tt(feats), tt(W), tt(nbr), tt(alpha), tt(beta), tt(res), True).cpu().numpy()
    assert np.array_equal(got, ref)


def test_spconv_vs_dense_torch_conv3d(dev):
    """independent check of the whole K4+K5 pair against torch's dense conv3d on a small grid (SubM semantics)."""
    rng = np.random.default_rng(0)
    shape = [9, 24, 20]
    occ = rng.random((2, *shape)) < 0.15
    idx = np.argwhere(occ).astype(np.int32)
    cin, cout = 16, 32
    feats = rng.standard_normal((len(idx), cin)).astype(np.float32)
    W = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    t = torch.from_numpy(idx).to(dev)
    table = ops.coord_table_build(t, shape, 2)
    nbr, _ = ops.rulebook_subm(t, shape, [3, 3, 3], table)
    got = ops.spconv_fwd(torch.from_numpy(feats).to(dev), torch.from_numpy(W).to(dev), nbr).cpu().numpy()
    dense = np.zeros((2, cin, *shape), np.float32)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feats
    w5 = torch.from_numpy(W.reshape(3, 3, 3, cin, cout)).permute(4, 3, 0, 1, 2).contiguous()
    full = torch.nn.functional.conv3d(torch.from_numpy(dense).double(), w5.double(), padding=1).numpy()
    ref = full[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_densify_exact(dev):
    idx = _level1(n=5000)
    oi, _, _, osh = O.rulebook_strided(idx, SHAPE1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    small = oi[(oi[:, 2] < 64) & (oi[:, 3] < 48)]
    shape = [osh[0], 64, 48]
    rng = np.random.default_rng(1)
    f = rng.standard_normal((len(small), 16)).astype(np.float32)
    ref = O.densify(f, small, 1, shape)
    got = ops.densify(torch.from_numpy(f).to(dev), torch.from_numpy(small).to(dev), 1, shape)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    assert got.view(1, 16 * shape[0], 64, 48).shape == (1, 16 * shape[0], 64, 48)


def _sort_key(idx, shape):
    """(b, y, x, z) lexicographic key = the bitmap cell index."""
    D, H, W = shape
    i = idx.astype(np.int64)
    return ((i[:, 0] * H + i[:, 2]) * W + i[:, 3]) * D + i[:, 1]


@pytest.mark.parametrize("batch", [1, 2])
def test_bitmap_rulebook_chain_matches_oracle(dev, batch):
    """Bitmap-rank rulebooks down the four strided levels of the nuScenes encoder: every active set comes out sorted by
    (b, y, x, z); compared with the oracle (i) as the order-free canonical pair sets of SURVEY 8a and (ii) exactly,
    after renumbering the oracle's rows into the same sorted order."""
    idx0 = _level1(batch=batch)
    shape = SHAPE1
    lvl, order, sidx = ops.bitmap_build(torch.from_numpy(idx0).to(dev), shape, batch)
    order = order.cpu().numpy()
    want_order = np.argsort(_sort_key(idx0, shape), kind="stable")
    np.testing.assert_array_equal(order, want_order)
    np.testing.assert_array_equal(sidx.cpu().numpy(), idx0[want_order])
    idx = idx0[want_order]
    # SubM on the sorted rows: identical to the oracle run on the same (sorted) rows
    nbr, cnt = O.rulebook_subm(idx, shape, [3, 3, 3])
    gn, gc = ops.rulebook_subm_bitmap(sidx, lvl, [3, 3, 3])
    np.testing.assert_array_equal(gn.cpu().numpy(), nbr)
    np.testing.assert_array_equal(gc.cpu().numpy(), cnt)
    specs = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
             ([3, 1, 1], [2, 1, 1], [0, 0, 0])]
    g_idx = sidx
    for ks, st, pd in specs:
        oi, onbr, ocnt, osh = O.rulebook_strided(idx, shape, ks, st, pd)
        gi, gn, gc, out_lvl, gsh = ops.rulebook_strided_bitmap(g_idx, lvl, ks, st, pd)
        assert gsh == osh
        gi_np = gi.cpu().numpy()
        key = _sort_key(gi_np, osh)
        assert np.all(np.diff(key) > 0)                       # sorted, distinct
        # the oracle's outputs are in first-seen order: renumber them into sorted order and compare exactly
        perm = np.argsort(_sort_key(oi, osh), kind="stable")
        np.testing.assert_array_equal(gi_np, oi[perm])
        np.testing.assert_array_equal(gn.cpu().numpy(), onbr[:, perm])
        np.testing.assert_array_equal(gc.cpu().numpy(), ocnt)
        for a, b in zip(_canon(gn.cpu().numpy(), idx, gi_np), _canon(onbr, idx, oi)):
            np.testing.assert_array_equal(a, b)
        # SubM on the new level through its bitmap
        sn, sc = O.rulebook_subm(gi_np, osh, [3, 3, 3])
        gsn, gsc = ops.rulebook_subm_bitmap(gi, out_lvl, [3, 3, 3])
        np.testing.assert_array_equal(gsn.cpu().numpy(), sn)
        np.testing.assert_array_equal(gsc.cpu().numpy(), sc)
        idx, shape, g_idx, lvl = gi_np, osh, gi, out_lvl


def test_bitmap_rulebook_empty_and_corner(dev):
    e = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    lvl, order, sidx = ops.bitmap_build(e, SHAPE1, 1)
    gn, gc = ops.rulebook_subm_bitmap(sidx, lvl, [3, 3, 3])
    assert gn.shape == (27, 0) and int(gc.sum()) == 0
    gi, gn, gc, _, _ = ops.rulebook_strided_bitmap(sidx, lvl, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert gi.shape[0] == 0 and gn.shape == (27, 0)
    one = np.array([[0, 40, 1471, 1471], [0, 0, 0, 0]], np.int32)  # far and near corner voxels
    lvl, order, sidx = ops.bitmap_build(torch.from_numpy(one).to(dev), SHAPE1, 1)
    np.testing.assert_array_equal(order.cpu().numpy(), [1, 0])
    oi, nbr, cnt, osh = O.rulebook_strided(one[[1, 0]], SHAPE1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    gi, gn, gc, _, _ = ops.rulebook_strided_bitmap(sidx, lvl, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    perm = np.argsort(_sort_key(oi, osh), kind="stable")
    np.testing.assert_array_equal(gi.cpu().numpy(), oi[perm])
    np.testing.assert_array_equal(gn.cpu().numpy(), nbr[:, perm])


@pytest.mark.parametrize("n,live", [(6000, None), (6000, 4100), (700, None), (3000, 0)])
def test_spconv_order_plan_is_a_sorted_permutation(dev, n, live):
    """srf_spconv_order_build: order = a permutation of the live rows (padding -1) that keeps every row inside its window of 1024, the
    plan's rulebook = the rulebook's columns in that order, and inside a window the rows are grouped by the key of their offset mask;
    the 32-channel convolution with the plan equals the one without, bit for bit, on a padded (static-shape) level too"""
    idx = _level1(n=n)
    nbr, _ = O.rulebook_subm(idx, SHAPE1, [3, 3, 3])
    A = nbr.shape[1]
    cap = A + 300                                    # capacity > live rows: the static-shape form
    nb = np.full((27, cap), -1, np.int32)
    nb[:, :A] = nbr
    n_live = A if live is None else min(live, A)
    t_nbr = torch.from_numpy(nb).to(dev)
    rows_dev = torch.tensor([n_live], dtype=torch.int32, device=dev)
    plan = ops.spconv_order(t_nbr, rows_dev).cpu().numpy()
    a_pad = (cap + 1023) // 1024 * 1024
    assert plan.shape[0] == a_pad * 28
    order, snbr = plan[:a_pad], plan[a_pad:].reshape(27, a_pad)
    livepos = order >= 0
    assert np.array_equal(np.sort(order[livepos]), np.arange(n_live))
    pos = np.nonzero(livepos)[0]
    assert np.array_equal(pos // 1024, order[livepos] // 1024)          # rows stay inside their window
    assert np.array_equal(snbr[:, livepos], nb[:, order[livepos]]) and (snbr[:, ~livepos] == -1).all()
    mask = (nb[:, :n_live] >= 0)
    m = (mask * (1 << np.arange(27))[:, None]).sum(0)
    key = ((((m & 0x1ff) != 0) * 1 + (((m >> 18) & 0x1ff) != 0) * 2) << 9) | ((m >> 9) & 0x1ff)
    for w in range((n_live + 1023) // 1024):
        inw = pos[(pos // 1024) == w]
        k = key[order[inw]]
        assert (np.diff(k) >= 0).all()                                     # grouped by key inside the window
    if n_live == 0:
        return
    rng = np.random.default_rng(n)
    for cin in (16, 32):
        feats = rng.standard_normal((A, cin)).astype(np.float32)
        W = (rng.standard_normal((27, cin, 32)) / np.sqrt(cin * 3)).astype(np.float32)
        res = rng.standard_normal((cap, 32)).astype(np.float32)
        tt = lambda x: torch.from_numpy(x).to(dev)
        packed = ops.pack_spconv_weights(tt(W))
        a = ops.spconv_fwd(tt(feats), tt(W), t_nbr, None, None, tt(res), True, packed=packed, rows_dev=rows_dev).cpu().numpy()
        b = ops.spconv_fwd(tt(feats), tt(W), t_nbr, None, None, tt(res), True, packed=packed, rows_dev=rows_dev,
                           tiles=ops.spconv_order(t_nbr, rows_dev)).cpu().numpy()
        assert np.array_equal(a[:n_live], b[:n_live])
