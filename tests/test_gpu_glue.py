"""Round 4, the small launches between the big kernels: each fused form against the torch ops it replaces
(srfdet_head.py:496-561 proposal generator, :957 centre sigmoid, :1002-1006 + :1246-1271 decode)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    return torch.device("cuda:0")


def test_decode_boxes_matches_the_torch_decode(dev):
    from srfdet3d_amd import ops
    from srfdet3d_amd.plugin.bbox_util import denormalize_bbox
    g = torch.Generator().manual_seed(3)
    pc_range = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    for Dd in (10, 8):
        logits = (torch.randn(2, 200, 10, generator=g) * 3).to(dev)
        pred = torch.randn(2, 200, Dd, generator=g)
        pred[..., :3] = torch.rand(2, 200, 3, generator=g)
        pred = pred.to(dev)
        scores, boxes = ops.decode_boxes(logits, pred, pc_range)
        lo = torch.tensor(pc_range[:3], device=dev)
        ext = torch.tensor(pc_range[3:], device=dev) - lo
        m = pred.clone()
        m[..., :3] = m[..., :3] * ext + lo          # the end of `forward`
        ref = denormalize_bbox(m, pc_range)          # SRFDetHead.decode
        ref[..., 2] = ref[..., 2] - ref[..., 5] * 0.5
        assert boxes.shape == ref.shape == (2, 200, Dd - 1)
        np.testing.assert_allclose(boxes.cpu().numpy(), ref.cpu().numpy(), rtol=2e-7, atol=1e-6)
        np.testing.assert_allclose(scores.cpu().numpy(), torch.sigmoid(logits).cpu().numpy(), rtol=2e-7, atol=1e-7)
        # centres: multiply, then add -- the same two roundings as torch
        assert torch.equal(boxes[..., :2], m[..., :2])


def test_dwconv_cat_is_copy_plus_dwconv(dev):
    from srfdet3d_amd import ops
    g = torch.Generator().manual_seed(5)
    for (N, H, W, C, Cs) in ((1, 46, 46, 256, 128), (6, 29, 51, 128, 64), (2, 7, 5, 8, 4)):
        x = torch.randn(N, H, W, C, generator=g).to(dev)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        side = torch.randn(N, Ho, Wo, Cs, generator=g).to(dev)
        w = torch.randn(C, 1, 3, 3, generator=g).to(dev)
        scale, shift = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev)
        ref = torch.empty(N, Ho, Wo, Cs + C, device=dev)
        ref[..., :Cs].copy_(side)
        ops.nhwc_dwconv3x3s2(x, w, scale, shift, True, out=ref[..., Cs:])
        out = torch.full((N, Ho, Wo, Cs + C), float("nan"), device=dev)
        got = ops.nhwc_dwconv3x3s2_cat(x, w, scale, shift, True, side, out)
        assert got.data_ptr() == out.data_ptr() and torch.equal(out, ref)
    # the level itself may be a channel slice of a wider buffer
    wide = torch.randn(1, 4, 4, 24, generator=g).to(dev)
    x = torch.randn(1, 8, 8, 8, generator=g).to(dev)
    w = torch.randn(8, 1, 3, 3, generator=g).to(dev)
    out = torch.empty(1, 4, 4, 16, device=dev)
    ops.nhwc_dwconv3x3s2_cat(x, w, None, None, False, wide[..., 8:16], out)
    assert torch.equal(out[..., :8], wide[..., 8:16])
    assert torch.equal(out[..., 8:], ops.nhwc_dwconv3x3s2(x, w))


def test_pool_sum_is_interpolate_camera_sum_channel_sum(dev):
    from srfdet3d_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7)
    # the BEV form: plain channel sum, 23 x 23 = 529 -> rows padded to 532 with zeros
    x = torch.randn(2, 23, 23, 512, generator=g).to(dev)
    out = ops.nhwc_pool_sum(x)
    assert out.shape == (2, 532) and torch.all(out[:, 529:] == 0)
    ref = x.double().sum(-1).flatten(1)
    np.testing.assert_allclose(out[:, :529].cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)
    # the camera form: nearest resize 29 x 50 -> 30 x 30 (and 30 x 15), six cameras summed
    for size in ((30, 30), (30, 15), (8, 64)):
        x = torch.randn(12, 29, 50, 64, generator=g).to(dev)
        out = ops.nhwc_pool_sum(x, n_cam=6, size=size)
        nchw = x.permute(0, 3, 1, 2)
        r = F.interpolate(nchw, list(size))                      # srfdet_head.py:550
        r = r.view(2, 6, 64, *size).double().sum(dim=1).sum(dim=1).flatten(1)
        assert out.shape[1] % 4 == 0 and out.shape[1] >= size[0] * size[1]
        np.testing.assert_allclose(out[:, :size[0] * size[1]].cpu().numpy(), r.cpu().numpy(), rtol=0, atol=2e-5)
    # repeatable bit for bit (fixed summation order)
    assert torch.equal(ops.nhwc_pool_sum(x, n_cam=6, size=(30, 30)), ops.nhwc_pool_sum(x, n_cam=6, size=(30, 30)))


def test_dpg_mix_matches_softmax_sums_sigmoid(dev):
    from srfdet3d_amd import ops
    g = torch.Generator().manual_seed(11)
    B, E, P, D, C = 2, 4, 200, 10, 128
    wl = torch.randn(B, E * P, generator=g).to(dev) * 2
    wi = torch.randn(B, E * P, generator=g).to(dev) * 2
    bw = torch.randn(E * P, D, generator=g).to(dev)
    fw = torch.randn(E * P, C, generator=g).to(dev)
    for second in (None, wi):
        boxes, feats = ops.dpg_mix(wl, second, bw, fw, E, P)
        w = wl.reshape(B, E, P)
        if second is not None:
            w = (w + second.reshape(B, E, P)) / 2
        w = w.softmax(1).unsqueeze(-1)                            # srfdet_head.py:554-556
        rb = (w * bw.view(1, E, P, -1)).sum(1)
        rf = (w * fw.view(1, E, P, -1)).sum(1)
        rb[..., :3] = rb[..., :3].sigmoid()                      # :957
        np.testing.assert_allclose(boxes.cpu().numpy(), rb.cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(feats.cpu().numpy(), rf.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_forward_decode_equals_decode_of_forward(dev):
    """the inference route (no per-stage box copies, fused proposal tail, one decode launch) against forward + decode"""
    from test_gpu_decoder import _gpu_head
    hd, feats = _gpu_head(dev)
    hd.eval()
    feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    with torch.no_grad():
        logits, boxes = hd(None, feats, None)
        ref_s, ref_b = hd.decode(logits, boxes)
        s, b = hd.forward_decode(None, feats, None)
        s2, b2 = hd.forward_decode(None, feats, None)
    assert torch.equal(s, s2) and torch.equal(b, b2)
    np.testing.assert_allclose(s.cpu().numpy(), ref_s.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b.cpu().numpy(), ref_b.cpu().numpy(), rtol=1e-6, atol=1e-6)
    # the proposals the stages start from: fused tail against the torch formulation
    with torch.no_grad():
        ib, pf = hd._get_init_proposals(None, feats)
        fb, ff = hd._stage_proposals(None, feats)
    ib = ib.clone()
    ib[..., :3] = ib[..., :3].sigmoid()
    np.testing.assert_allclose(fb.cpu().numpy(), ib.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ff.cpu().numpy(), pf.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_hard_voxelize_static_writes_its_own_padding(dev):
    from srfdet3d_amd import ops
    g = torch.Generator().manual_seed(13)
    vs, rng = [0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    n = 20000
    pts = torch.rand(n, 5, generator=g) * torch.tensor([100.0, 100.0, 7.0, 1.0, 1.0]) - torch.tensor([50.0, 50.0, 4.5, 0.0, 0.0])
    pts[15000:] = 1.0e6                                    # the out-of-range filler rows of a static frame
    pts = pts.to(dev)
    v, c, num, mean = ops.hard_voxelize(pts, vs, rng, 10, 120000, mean_features=5)
    M = v.shape[0]
    # poisoned allocator memory must not show through: fill the pool with NaNs / garbage first
    junk = torch.full((n * 10 * 5 + 8 * n,), float("nan"), device=dev)
    del junk
    sv, sc, snum, smean, vnum = ops.hard_voxelize(pts, vs, rng, 10, 120000, mean_features=5, static=True, batch_index=0)
    assert int(vnum.item()) == M and sv.shape[0] == n and sc.shape == (n, 4)
    assert torch.equal(sv[:M], v) and torch.equal(snum[:M], num) and torch.equal(smean[:M], mean)
    assert torch.equal(sc[:M, 1:], c) and torch.all(sc[:M, 0] == 0)
    assert torch.all(sc[M:] == -1) and torch.all(snum[M:] == 0) and torch.all(sv[M:] == 0) and torch.all(smean[M:] == 0)
    sc3 = ops.hard_voxelize(pts, vs, rng, 10, 120000, mean_features=5, static=True, batch_index=3)[1]
    assert torch.all(sc3[:M, 0] == 3) and torch.all(sc3[M:] == -1)


def test_padded_bitmap_levels_fill_their_own_tails(dev):
    """srf_bitmap_build_padded / srf_bitmap_strided_outputs_static write the slots past the live count themselves"""
    from srfdet3d_amd import ops
    g = torch.Generator().manual_seed(17)
    shape = [41, 128, 128]
    cells = torch.randperm(41 * 128 * 128, generator=g)[:5000]
    z, y, x = cells // (128 * 128), (cells // 128) % 128, cells % 128
    idx = torch.stack([torch.zeros_like(z), z, y, x], 1).int()
    padded = torch.cat([idx, torch.full((700, 4), -1, dtype=torch.int32)]).to(dev)
    lvl, order, sidx = ops.bitmap_build(idx.to(dev), shape, 1)
    torch.full((64 * 5700,), 12345, dtype=torch.int32, device=dev)  # dirty the pool
    plvl, porder, psidx = ops.bitmap_build(padded, shape, 1, padded=True)
    assert torch.equal(psidx[:5000], sidx) and torch.equal(porder[:5000], order)
    assert torch.all(psidx[5000:] == -1) and torch.all(porder[5000:] == 0)
    assert torch.equal(plvl.bitmap, lvl.bitmap)
    out_idx, nbr, counts, out_lvl, oshape = ops.rulebook_strided_bitmap(sidx, lvl, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    A_out = out_idx.shape[0]
    torch.full((64 * 5700,), 12345, dtype=torch.int32, device=dev)
    s_idx, s_nbr, _, s_lvl, s_shape, s_num = ops.rulebook_strided_bitmap(psidx, plvl, [3, 3, 3], [2, 2, 2], [1, 1, 1], out_capacity=A_out + 300)
    assert int(s_num.item()) == A_out and s_idx.shape[0] == A_out + 300
    assert torch.equal(s_idx[:A_out], out_idx) and torch.all(s_idx[A_out:] == -1)
    assert torch.equal(s_nbr[:, :A_out], nbr) and torch.all(s_nbr[:, A_out:] == -1)


def test_host_pack(dev):
    from srfdet3d_amd import ops
    a = torch.randn(3, 50, 11, device=dev)
    b = torch.tensor([[7, 12], [0, 3], [50, 16000000]], dtype=torch.int32, device=dev)
    c = torch.tensor([29871, 4, 0, 119], dtype=torch.int32, device=dev)
    out = ops.host_pack(a, b, c)
    ref = torch.cat([a.reshape(-1), b.reshape(-1).float(), c.float()])
    assert torch.equal(out, ref)


def test_densify_bev_is_dense_view_channels_last(dev):
    """ops.densify_bev (one pass from the bitmap, channels-last) against dense() + the (N, C*D, H, W) view"""
    from srfdet3d_amd import ops
    g = torch.Generator().manual_seed(19)
    for (B, D, H, W, C, A) in ((1, 2, 180, 180, 128, 9000), (2, 2, 45, 51, 64, 3000), (1, 3, 20, 24, 16, 700)):
        cells = torch.randperm(B * D * H * W, generator=g)[:A]
        b = cells // (D * H * W)
        r = cells % (D * H * W)
        z, y, x = r // (H * W), (r // W) % H, r % W
        idx = torch.stack([b, z, y, x], 1).int().to(dev)
        feats = torch.randn(A, C, generator=g).to(dev)
        lvl, order, sidx = ops.bitmap_build(idx, [D, H, W], B)
        sf = feats[order.long()]
        ref = ops.densify(sf, sidx, B, [D, H, W]).view(B, C * D, H, W)
        torch.full((B * C * D * H * W,), float("nan"), device=dev)   # dirty the pool: every element must be written
        got = ops.densify_bev(sf, lvl, B, [D, H, W])
        assert got.shape == ref.shape and got.stride(1) == 1 and torch.equal(got, ref)
    # a capacity-sized set with padding rows behind the live ones (static frames)
    pad = torch.cat([sidx, torch.full((100, 4), -1, dtype=torch.int32, device=dev)])
    padf = torch.cat([sf, torch.full((100, C), float("nan"), device=dev)])
    assert torch.equal(ops.densify_bev(padf, lvl, B, [D, H, W]), ref)


def test_ese_apply_equals_gate_then_affine(dev):
    """srf_ese_apply (gate GEMV + multiply + identity add in one launch) against srf_ese_gate followed by srf_nhwc_affine: same bits"""
    from srfdet3d_amd import ops
    g = torch.Generator().manual_seed(29)
    for (N, H, W, C, ident) in ((6, 29, 50, 1024, True), (2, 58, 100, 768, True), (3, 17, 23, 256, False), (1, 5, 3, 512, True), (1, 300, 9, 64, False)):
        wide = torch.randn(N, H, W, C + 64, generator=g).to(dev)
        x = wide[..., 32:32 + C] if C % 4 == 0 else wide[..., :C]        # a channel slice of a wider pixel-major buffer
        mean = torch.randn(N, C, generator=g).to(dev)
        w = (torch.randn(C, C, 1, 1, generator=g) * 0.05).to(dev)
        b = torch.randn(C, generator=g).to(dev)
        res = torch.randn(N, H, W, C, generator=g).to(dev) if ident else None
        gate = ops.ese_gate(mean, w, b)
        ref = ops.nhwc_affine(x, scale=gate, residual=res)
        out = torch.full((N, H, W, C), float("nan"), device=dev)
        got, gate2 = ops.ese_apply(x, mean, w, b, residual=res, out=out, want_gate=True)
        assert got.data_ptr() == out.data_ptr() and torch.equal(gate2, gate) and torch.equal(out, ref)
        # the definition in torch: hsigmoid(fc(mean)) gate, multiply, identity add
        tg = torch.nn.functional.hardsigmoid(mean @ w.view(C, C).t() + b)
        tref = x * tg.view(N, 1, 1, C) + (res if ident else 0)
        np.testing.assert_allclose(out.cpu().numpy(), tref.cpu().numpy(), rtol=1e-5, atol=1e-5)
