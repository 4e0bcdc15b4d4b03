"""Gradients of the sparse half (csrc/spconv_bwd.hip + the autograd Functions of ops.py) against torch autograd on a
densified small grid: SubM and strided sparse convolutions (data and weight gradients, every channel pair of the
encoders), SparseConvTensor.dense(), DynamicScatter mean / max, and a LiDAR-only training step with the encoder unfrozen
(the reference trains it in every L-only config, tools/train.py:221-234)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from srfdet3d_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _dense_grads(idx, shape, feats, W, gout_rows, out_idx, out_shape, stride, pad, ksize):
    """float64 reference: conv3d on the densified grid; the loss is sum(out[active outputs] * gout_rows)."""
    B = int(idx[:, 0].max()) + 1
    cin, cout = W.shape[1], W.shape[2]
    f = torch.from_numpy(feats).double().requires_grad_(True)
    w = torch.from_numpy(W).double().requires_grad_(True)
    i = torch.from_numpy(idx).long()
    # channels-last dense grid so that an active site is one row: (B, D, H, W, C) -> (B, C, D, H, W)
    dense = torch.zeros(B, *shape, cin, dtype=torch.float64).index_put((i[:, 0], i[:, 1], i[:, 2], i[:, 3]), f).permute(0, 4, 1, 2, 3)
    w5 = w.view(*ksize, cin, cout).permute(4, 3, 0, 1, 2)
    full = F.conv3d(dense, w5, stride=stride, padding=pad)
    o = torch.from_numpy(out_idx).long()
    out = full[o[:, 0], :, o[:, 1], o[:, 2], o[:, 3]]
    (out * torch.from_numpy(gout_rows).double()).sum().backward()
    return out.detach().numpy(), f.grad.numpy(), w.grad.numpy()


@pytest.mark.parametrize("cin,cout", [(5, 16), (4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
def test_subm_conv_gradients(dev, cin, cout):
    rng = np.random.default_rng(cin * 131 + cout)
    shape = [7, 18, 16]
    idx = np.argwhere(rng.random((2, *shape)) < 0.2).astype(np.int32)
    feats = rng.standard_normal((len(idx), cin)).astype(np.float32)
    W = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    gout = rng.standard_normal((len(idx), cout)).astype(np.float32)
    t = torch.from_numpy(idx).to(dev)
    nbr, _ = ops.rulebook_subm(t, shape, [3, 3, 3], ops.coord_table_build(t, shape, 2))
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    w = torch.from_numpy(W).to(dev).requires_grad_(True)
    out = ops.spconv_fwd(f, w, nbr, subm=True)
    (out * torch.from_numpy(gout).to(dev)).sum().backward()
    ref_out, ref_gf, ref_gw = _dense_grads(idx, shape, feats, W, gout, idx, shape, 1, 1, (3, 3, 3))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out, rtol=1e-5, atol=1e-5)
    assert np.abs(f.grad.cpu().numpy() - ref_gf).max() <= 1e-4 * np.abs(ref_gf).max()
    assert np.abs(w.grad.cpu().numpy() - ref_gw).max() <= 1e-4 * np.abs(ref_gw).max()


@pytest.mark.parametrize("cin,cout,ksize,stride,pad", [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (64, 128, (3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                                       (128, 128, (3, 1, 1), (2, 1, 1), (0, 0, 0)), (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1))])
def test_strided_conv_gradients(dev, cin, cout, ksize, stride, pad):
    rng = np.random.default_rng(cin + 7 * cout)
    shape = [9, 20, 14]
    idx = np.argwhere(rng.random((2, *shape)) < 0.12).astype(np.int32)
    K = ksize[0] * ksize[1] * ksize[2]
    feats = rng.standard_normal((len(idx), cin)).astype(np.float32)
    W = (rng.standard_normal((K, cin, cout)) * 0.1).astype(np.float32)
    t = torch.from_numpy(idx).to(dev)
    out_idx, nbr, _, _, oshape = ops.rulebook_strided(t, shape, 2, list(ksize), list(stride), list(pad))
    gout = rng.standard_normal((out_idx.shape[0], cout)).astype(np.float32)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    w = torch.from_numpy(W).to(dev).requires_grad_(True)
    out = ops.spconv_fwd(f, w, nbr, subm=False)
    (out * torch.from_numpy(gout).to(dev)).sum().backward()
    ref_out, ref_gf, ref_gw = _dense_grads(idx, shape, feats, W, gout, out_idx.cpu().numpy(), list(oshape), stride, pad, ksize)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out, rtol=1e-5, atol=1e-5)
    assert np.abs(f.grad.cpu().numpy() - ref_gf).max() <= 1e-4 * np.abs(ref_gf).max()
    assert np.abs(w.grad.cpu().numpy() - ref_gw).max() <= 1e-4 * np.abs(ref_gw).max()


def test_fused_epilogue_route_is_differentiable(dev):
    """A frozen (eval) BatchNorm folded into the layer, residual and ReLU, with gradients requested: the result equals the
    fused inference kernel and the gradients equal torch's on the same composition."""
    rng = np.random.default_rng(3)
    shape = [5, 12, 10]
    idx = np.argwhere(rng.random((1, *shape)) < 0.3).astype(np.int32)
    t = torch.from_numpy(idx).to(dev)
    nbr, _ = ops.rulebook_subm(t, shape, [3, 3, 3], ops.coord_table_build(t, shape, 1))
    A = len(idx)
    f = torch.from_numpy(rng.standard_normal((A, 32)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((27, 32, 32)) * 0.1).astype(np.float32)).to(dev)
    alpha = torch.rand(32, device=dev) + 0.5
    beta = torch.randn(32, device=dev)
    res = torch.randn(A, 32, device=dev)
    with torch.no_grad():
        want = ops.spconv_fwd(f, w, nbr, alpha, beta, res, True, subm=True)
    f2, w2 = f.clone().requires_grad_(True), w.clone().requires_grad_(True)
    got = ops.spconv_fwd(f2, w2, nbr, alpha, beta, res, True, subm=True)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    got.sum().backward()
    assert f2.grad is not None and w2.grad is not None and torch.isfinite(w2.grad).all() and w2.grad.abs().sum() > 0


def test_densify_and_scatter_gradients(dev):
    rng = np.random.default_rng(5)
    shape = [4, 10, 12]
    idx = np.argwhere(rng.random((2, *shape)) < 0.25).astype(np.int32)
    A = len(idx)
    f = torch.from_numpy(rng.standard_normal((A, 16)).astype(np.float32)).to(dev).requires_grad_(True)
    dense = ops.densify(f, torch.from_numpy(idx).to(dev), 2, shape)
    g = torch.randn_like(dense)
    (dense * g).sum().backward()
    i = torch.from_numpy(idx).long()
    np.testing.assert_array_equal(f.grad.cpu().numpy(), g.cpu()[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].numpy())
    # DynamicScatter mean / max: points of 40 voxels, several per voxel, against index_add / amax autograd
    from srfdet3d_amd.voxel_layer import DynamicScatter
    n = 300
    cells = rng.integers(0, 40, n)
    coors = np.stack([np.zeros(n), cells // 20, (cells // 5) % 4, cells % 5], 1).astype(np.int32)
    coors[::17] = -1    # dropped points
    pts = rng.standard_normal((n, 6)).astype(np.float32)
    sc = DynamicScatter([0.1, 0.1, 0.1], [0, 0, 0, 0.5, 0.4, 0.2], True)
    vm = sc.voxel_map(torch.from_numpy(coors).to(dev))
    for mode in ("mean", "max"):
        x = torch.from_numpy(pts).to(dev).requires_grad_(True)
        out = vm.reduce(x, mode)
        gv = torch.randn_like(out)
        (out * gv).sum().backward()
        xr = torch.from_numpy(pts).double().requires_grad_(True)
        p2v = vm.point2voxel.cpu().long()
        ok = p2v >= 0
        if mode == "mean":
            ref = torch.zeros(vm.M, 6, dtype=torch.float64).index_add(0, p2v[ok], xr[ok])
            ref = ref / torch.bincount(p2v[ok], minlength=vm.M).unsqueeze(1)
        else:
            ref = torch.full((vm.M, 6), -float("inf"), dtype=torch.float64).scatter_reduce(0, p2v[ok].unsqueeze(1).expand(-1, 6), xr[ok], "amax")
        (ref * gv.cpu().double()).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_lidar_only_training_step_trains_the_sparse_encoder(dev):
    """srfdet_voxel_nusc_L with the LiDAR branch UNFROZEN (the reference's L-only training): forward_train -> loss_ota ->
    backward -> AdamW step; every sparse-encoder weight receives a finite, non-zero gradient.  A second model whose first
    sparse block is replaced by the dense float64 equivalent is not needed: the per-layer gradients are checked against
    conv3d autograd above, here the chain through BatchNorm1d (train mode), residuals, dense() and SECOND is exercised."""
    from srfdet3d_amd import synthetic as S, workloads
    from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
    torch.manual_seed(0)
    model = workloads.build("srfdet_voxel_nusc_L", 64, train=True).to(dev).train()
    pts = torch.from_numpy(S.nuscenes_sweep(2000, 6000)).to(dev)
    g = torch.Generator().manual_seed(2)
    n = 8
    b = torch.cat([torch.rand(n, 2, generator=g) * 60 - 30, torch.rand(n, 1, generator=g) * 2 - 3, torch.rand(n, 3, generator=g) * 3 + 1,
                   torch.rand(n, 1, generator=g) * 6 - 3, torch.randn(n, 2, generator=g)], 1)
    gtb = LiDARInstance3DBoxes(b.to(dev), box_dim=9)
    gtl = torch.randint(0, 10, (n,), generator=g).to(dev)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    losses = model(return_loss=True, img=None, points=[pts], img_metas=[dict(box_type_3d=LiDARInstance3DBoxes)], gt_bboxes_3d=[gtb],
                   gt_labels_3d=[gtl])
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    enc = dict(model.pts_middle_encoder.named_parameters())
    convs = [k for k in enc if k.endswith("weight") and enc[k].dim() >= 3]
    assert len(convs) == 21
    for k in convs:
        gr = enc[k].grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0, k
    before = enc["conv_input.0.weight"].detach().clone()
    opt.step()
    assert not torch.equal(before, enc["conv_input.0.weight"])
