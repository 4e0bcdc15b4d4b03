"""GPU parity: K7 RoIAlign / SingleRoIExtractor gather and the fused box -> RoI geometry."""
import numpy as np
import pytest
import torch

import detgen
from oracle import oracle as O
from srfdet3d_amd import ops, synthetic as S

pytestmark = pytest.mark.gpu
GOLD = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "decoder_nusc.npz"))


def _pyramid(rng, C=128, sizes=(184, 92, 46, 23), N=1):
    return [rng.standard_normal((N, C, s, s)).astype(np.float32) for s in sizes]


def _rois(rng, R, size=1472, N=1):
    c = rng.uniform(0, size, (R, 2))
    wh = np.exp(rng.uniform(np.log(4), np.log(900), (R, 2)))
    b = rng.integers(0, N, (R, 1))
    r = np.concatenate([b, c - wh / 2, c + wh / 2], 1).astype(np.float32)
    r[:8, 1:] += 2000  # entirely outside the map: samples read zeros
    r[8:12, 3:] = r[8:12, 1:3]  # zero-area RoIs
    return r


def test_roi_extract_matches_oracle(dev):
    rng = np.random.default_rng(0)
    feats = _pyramid(rng, N=2)
    rois = _rois(rng, 300, N=2)
    ref, lvl = O.roi_extract(feats, rois, [8, 16, 32, 64])
    tf = [torch.from_numpy(f).to(dev) for f in feats]
    got, glv = ops.roi_extract(tf, torch.from_numpy(rois).to(dev), [8, 16, 32, 64], return_levels=True)
    np.testing.assert_array_equal(glv.cpu().numpy(), lvl)
    assert set(np.unique(lvl)) == {0, 1, 2, 3}
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    # channels-last maps and the bin-major output layout give the same numbers
    tcl = [f.contiguous(memory_format=torch.channels_last) for f in tf]
    got2 = ops.roi_extract(tcl, torch.from_numpy(rois).to(dev), [8, 16, 32, 64], bin_major=True)
    np.testing.assert_array_equal(got2.permute(0, 2, 1).reshape(ref.shape).cpu().numpy(), ref)


def test_roi_extract_accumulate_sums_cameras(dev):
    rng = np.random.default_rng(1)
    feats = [rng.standard_normal((6, 32, s, 2 * s)).astype(np.float32) for s in (58, 29, 15, 8)]
    P = 40
    rois = np.concatenate([_rois(rng, P, size=400) for _ in range(6)], 0)
    rois[:, 0] = np.repeat(np.arange(6), P)
    ref, _ = O.roi_extract(feats, rois, [4, 8, 16, 32])
    ref = ref.reshape(6, P, 32, 7, 7)
    acc = ref[0].copy()
    for c in range(1, 6):
        acc = acc + ref[c]
    tf = [torch.from_numpy(f).to(dev) for f in feats]
    out = None
    for c in range(6):
        r = torch.from_numpy(rois[c * P:(c + 1) * P]).to(dev)
        out = ops.roi_extract(tf, r, [4, 8, 16, 32], out=out, accumulate=c > 0)
    np.testing.assert_array_equal(out.cpu().numpy(), acc)


def test_box_rois_match_reference_fixture(dev):
    """geometry vs the RoIs the reference's own code asked its pooler for (tests/golden/make_fixtures.py)."""
    from make_fixtures import det_boxes, NUSC_RANGE, NUSC_VOXEL
    P = 48
    bx = torch.from_numpy(det_boxes("lstage.boxes", P)).to(dev)
    rb, _ = ops.box_rois(bx, NUSC_RANGE, NUSC_VOXEL, mutate_centres=True)
    # BEV pixel coordinates live in [0, 1472]; float32 spacing there is 1.2e-4
    np.testing.assert_allclose(rb.cpu().numpy(), GOLD["lstage.rois"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(bx.cpu().numpy(), GOLD["lstage.boxes_after"], rtol=1e-6, atol=1e-5)

    bx = torch.from_numpy(det_boxes("fstage.boxes", P)).to(dev)
    l2i = torch.from_numpy(S.camera_rig()[None]).to(dev)
    rb, ri = ops.box_rois(bx, NUSC_RANGE, NUSC_VOXEL, mutate_centres=False, lidar2img=l2i)
    np.testing.assert_allclose(rb.cpu().numpy(), GOLD["fstage.rois_lidar"], rtol=0, atol=2e-3)
    ref = GOLD["fstage.rois_img"]
    got = ri.cpu().numpy()
    np.testing.assert_array_equal(got[:, 0], ref[:, 0])
    # projected corners reach 1e7 px when a corner sits near the camera plane: compare relatively
    np.testing.assert_allclose(got[:, 1:], ref[:, 1:], rtol=2e-4, atol=5e-2)


def test_roi_extract_sum_is_the_ordered_camera_sum_into_a_strided_slice(dev):
    """srf_roi_extract_sum (the image gather of a fusion stage, srfdet_head.py:2543-2562): row r = gather(roi[0 R + r]) + gather(roi[1 R + r])
    + ... added in camera order, written into a channel slice of the (R, S, 2 C) operand of `output_fused_proj`; the plain gather into
    the other slice.  Against the oracle's per-RoI gathers summed in the same order: exact."""
    rng = np.random.default_rng(3)
    n_cam, R, C = 6, 40, 128
    feats = _pyramid(rng, C=C, sizes=(58, 29, 15, 8), N=n_cam)
    rois = _rois(rng, n_cam * R, size=460, N=n_cam)
    ref, _ = O.roi_extract(feats, rois, [8, 16, 32, 64])                    # (n_cam R, C, 7, 7)
    want = ref[:R].copy()
    for s in range(1, n_cam):
        want = want + ref[s * R:(s + 1) * R]                                 # sequential float32 adds, camera order
    want = want.reshape(R, C, 49).transpose(0, 2, 1)                         # bin-major (R, S, C)
    tcl = [torch.from_numpy(f).to(dev).contiguous(memory_format=torch.channels_last) for f in feats]
    buf = torch.full((R, 49, 2 * C), 7.0, device=dev)
    out = ops.roi_extract(tcl, torch.from_numpy(rois).to(dev), [8, 16, 32, 64], out=buf[..., :C], bin_major=True, n_sum=n_cam)
    assert out.data_ptr() == buf.data_ptr()
    np.testing.assert_array_equal(buf[..., :C].cpu().numpy(), want)
    assert torch.all(buf[..., C:] == 7.0)                                    # the other half is untouched
    # the plain gather into the right half of the same buffer
    bev = _pyramid(rng, C=C, sizes=(46, 23, 12, 6), N=1)
    rb = _rois(rng, R, size=368, N=1)
    refb, _ = O.roi_extract(bev, rb, [8, 16, 32, 64])
    tb = [torch.from_numpy(f).to(dev).contiguous(memory_format=torch.channels_last) for f in bev]
    ops.roi_extract(tb, torch.from_numpy(rb).to(dev), [8, 16, 32, 64], out=buf[..., C:], bin_major=True)
    np.testing.assert_array_equal(buf[..., C:].cpu().numpy(), refb.reshape(R, C, 49).transpose(0, 2, 1))
    np.testing.assert_array_equal(buf[..., :C].cpu().numpy(), want)
    with pytest.raises(ValueError):
        ops.roi_extract(tcl, torch.from_numpy(rois[:7]).to(dev), [8, 16, 32, 64], n_sum=n_cam)
