"""Host-side decoder arithmetic against fixtures produced by the REFERENCE's own Python
(tests/golden/make_fixtures.py -> tests/golden/decoder_nusc.npz).  CPU torch; no HIP op is involved here --
the HIP geometry / gather in front of this arithmetic is covered by tests/test_gpu_decoder.py.

Because inputs and weights are regenerated from (name, shape) on both sides, a pass also proves that every
parameter of the stage modules has the same name and shape as in the reference (checkpoint compatibility)."""
import os

import numpy as np
import torch

import detgen
from make_fixtures import NUSC_RANGE, STAGE_KW, det_boxes
from srfdet3d_amd.plugin import bbox_util, heads

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "decoder_nusc.npz"))
P = 48
t = torch.from_numpy


def _abs_boxes():
    b = det_boxes("boxutil.boxes", P)
    b[..., :3] = b[..., :3] * 100.0 - 50.0
    return b


def test_box_utilities_match_reference():
    b = _abs_boxes()
    got = bbox_util.boxes3d_to_corners3d(t(b[..., :8].copy()), bottom_center=False, ry=False).numpy()
    np.testing.assert_allclose(got, GOLD["corners3d"], rtol=1e-6, atol=2e-5)
    den = bbox_util.denormalize_bbox(t(b[0].copy()), NUSC_RANGE)
    np.testing.assert_allclose(den.numpy(), GOLD["denormalize"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(bbox_util.normalize_bbox(den, NUSC_RANGE).numpy(), GOLD["normalize"], rtol=1e-6, atol=1e-6)


def test_dynamic_conv_matches_reference():
    dc = heads.DynamicConv(128, dynamic_dim=32, dynamic_num=2, pooler_resolution=7).eval()
    detgen.load_det_params(dc, "dynconv.")
    with torch.no_grad():
        got = dc(t(detgen.det("dynconv.prop", (1, P, 128))), t(detgen.det("dynconv.roi", (49, P, 128))))
    np.testing.assert_allclose(got.numpy(), GOLD["dynconv"], rtol=1e-4, atol=1e-5)


def _bin_major(x):
    return x.flatten(2).permute(0, 2, 1).contiguous()


def test_lidar_stage_arithmetic_matches_reference():
    st = heads.SingleSRFDetHeadLiDAR(**STAGE_KW).eval()
    detgen.load_det_params(st, "lstage.")
    roi = _bin_major(t(detgen.det("lstage.roi_feats", (P, 128, 7, 7))))
    with torch.no_grad():
        logits, pred, obj = st._refine(roi, t(GOLD["lstage.boxes_after"].copy()), t(detgen.det("lstage.prop", (1, P, 128))), 1, P)
    np.testing.assert_allclose(logits.numpy(), GOLD["lstage.logits"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pred.numpy(), GOLD["lstage.pred"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(obj.numpy(), GOLD["lstage.obj"], rtol=1e-4, atol=1e-4)


def test_apply_deltas_with_clamp_matches_reference():
    st = heads.SingleSRFDetHeadLiDAR(**STAGE_KW).eval()
    deltas = detgen.det("deltas.d", (P, 10), scale=0.5)
    deltas[:4, 3:6] = 12.0
    got = st.apply_deltas_lidar(t(deltas), t(_abs_boxes()[0].copy()))
    np.testing.assert_allclose(got.numpy(), GOLD["apply_deltas"], rtol=1e-5, atol=1e-5)


def test_fusion_stage_arithmetic_matches_reference():
    fs = heads.SingleSRFDetHead(use_fusion=True, **STAGE_KW).eval()
    detgen.load_det_params(fs, "fstage.")
    img = _bin_major(t(detgen.det("fstage.roi_img", (6 * P, 128, 7, 7)))).view(6, P, 49, 128).sum(0)
    pts = _bin_major(t(detgen.det("fstage.roi_lidar", (P, 128, 7, 7))))
    boxes = det_boxes("fstage.boxes", P)
    lo, hi = np.array(NUSC_RANGE[:3], np.float32), np.array(NUSC_RANGE[3:], np.float32)
    boxes[..., :3] = boxes[..., :3] * (hi - lo) + lo
    with torch.no_grad():
        roi = fs.output_fused_proj(torch.cat((img, pts), dim=-1))
        logits, pred, obj = fs._refine(roi, t(boxes), t(detgen.det("fstage.prop", (1, P, 128))), 1, P)
    np.testing.assert_allclose(logits.numpy(), GOLD["fstage.logits"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(pred.numpy(), GOLD["fstage.pred"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(obj.numpy(), GOLD["fstage.obj"], rtol=1e-4, atol=2e-4)


def _nusc_head(num_proposals):
    import srfdet3d_amd as S
    from srfdet3d_amd import workloads
    m = workloads.model_cfg("srfdet_voxel_nusc_L")
    hc = dict(m.bbox_head)
    hc.update(num_proposals=num_proposals, train_cfg=None, test_cfg=m.test_cfg, use_img=False)
    return S.compat.build_head(hc).eval()


def test_decode_matches_reference():
    hd = _nusc_head(8)
    b = _abs_boxes()
    with torch.no_grad():
        scores, boxes = hd.decode(t(detgen.det("decode.logits", (5, 1, P, 10))), t(b[None].repeat(5, 0).copy()))
    np.testing.assert_allclose(boxes[0].numpy(), GOLD["decode.boxes"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(scores[0].numpy(), GOLD["decode.scores"][:, :10], rtol=1e-6, atol=1e-7)


def test_dpg_proposals_match_reference():
    hd = _nusc_head(32)
    detgen.load_det_params(hd, "head.")
    feats = [t(detgen.det(f"head.feat{i}", (1, 128, s, s), scale=0.5)) for i, s in enumerate((184, 92, 46, 23))]
    with torch.no_grad():
        boxes, pf = hd._get_init_proposals(None, feats)
    np.testing.assert_allclose(boxes.numpy(), GOLD["head.init_boxes"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pf.numpy(), GOLD["head.init_feats"], rtol=1e-4, atol=1e-5)
