"""The HIP paths against the round-2 fixtures produced by the REFERENCE's own Python (tests/golden/extra_r2.npz):
DynamicVFECustom on the HIP scatter kernels, SECONDCustom on the GPU convolution path, the decoder stage at the KITTI
arguments (C = 256, d = 64, ff = 1024, P = 100) and the nuScenes stage at P = 200 / 900 through the HIP geometry and stage
kernels, the `apply_deltas` fixture WITH the clamp branch through srf_apply_deltas and srf_stage_tail, the OTA assigner +
loss_ota on the HIP rotated IoU, and an LC training step at two frames per GPU with the config's own np = 900."""
import os

import numpy as np
import pytest
import torch

import detgen
import make_fixtures as mf
import make_fixtures_r2 as mf2
from srfdet3d_amd import ops
from srfdet3d_amd.plugin import backbones, heads, training, voxel_encoders
from test_fixtures_r2 import check_loss_ota, check_ota, ota_inputs

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "extra_r2.npz"))
GOLD1 = np.load(os.path.join(os.path.dirname(__file__), "golden", "decoder_nusc.npz"))
t = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag,kw,nf", [
    ("vfe_kitti", dict(in_channels=4, feat_channels=[4], voxel_size=mf2.KITTI_VOXEL, point_cloud_range=mf2.KITTI_RANGE), 4),
    ("vfe_waymo", dict(in_channels=5, feat_channels=[5, 5], voxel_size=mf2.WAYMO_VOXEL, point_cloud_range=mf2.WAYMO_RANGE), 5)])
def test_dynamic_vfe_matches_reference(dev, tag, kw, nf):
    """DynamicVFELayer + DynamicVFECustom.forward (utils.py:30-45, voxel_encoder.py:162-240): per-point coordinates from the
    HIP dynamic voxelizer equal the fixture's, the sorted unique voxels are equal, the voxel features agree to 1e-4.  The
    reference run used a restatement of mmcv's DynamicScatter (the unpinned part)."""
    enc = voxel_encoders.DynamicVFECustom(with_cluster_center=True, with_voxel_center=True, with_distance=False,
                                          norm_cfg=dict(type="naiveSyncBN1dCustom", eps=1e-3, momentum=0.01), **kw).eval()
    detgen.load_det_params(enc, tag + ".")
    enc = enc.to(dev)
    pts, coors = mf2.vfe_points(tag, kw["point_cloud_range"], kw["voxel_size"], nf)
    np.testing.assert_array_equal(coors, GOLD[tag + ".coors_in"])
    p = t(pts).to(dev)
    zyx = ops.dynamic_voxelize(p, kw["voxel_size"], kw["point_cloud_range"])
    np.testing.assert_array_equal(zyx.cpu().numpy(), coors[:, 1:])
    c = torch.cat([t(coors[:, :1]).to(dev), zyx], dim=1)
    with torch.no_grad():
        vf, vc = enc(p, c)
    np.testing.assert_array_equal(vc.cpu().numpy(), GOLD[tag + ".voxel_coors"])
    np.testing.assert_allclose(vf.cpu().numpy(), GOLD[tag + ".voxel_feats"], rtol=1e-4, atol=1e-4)


def test_second_custom_matches_reference(dev):
    """SECONDCustom.forward (second_custom.py:78-91), KITTI arguments, through the GPU convolution path of this repo."""
    net = backbones.SECONDCustom(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2]).eval()
    detgen.load_det_params(net, "second.")
    net = net.to(dev)
    with torch.no_grad():
        o = net(t(detgen.det("second.x", (1, 256, 24, 20), scale=0.5)).to(dev))
    for got, key in zip(o, ("second.out0", "second.out1")):
        want = GOLD[key]
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-4 * np.abs(want).max()


class _FixedPooler:
    """Stands where the RoI extractor is called (mmdet call surface): returns the fixture's RoI features, keeps the RoIs."""
    num_inputs = 4

    def __init__(self, out):
        self.out, self.rois = out, None

    def __call__(self, feats, rois):
        self.rois = rois.detach().clone()
        return self.out.clone()


@pytest.mark.parametrize("tag,kw,P,C,D", [("kstage", mf2.KSTAGE_KW, 100, 256, 8), ("lstage200", mf.STAGE_KW, 200, 128, 10),
                                          ("lstage900", mf.STAGE_KW, 900, 128, 10)])
def test_stage_on_gpu_matches_reference(dev, tag, kw, P, C, D):
    """One decoder stage on the GPU (HIP box -> RoI geometry, HIP stage kernels) against the reference's stage at the KITTI
    arguments and at the measurement / config proposal counts: RoIs, box parameters within 1e-4, logits, object features."""
    st = heads.SingleSRFDetHeadLiDAR(**kw).eval()
    detgen.load_det_params(st, tag + ".")
    st = st.to(dev)
    bx = t(mf.det_boxes(tag + ".boxes", P)[..., :D].copy()).to(dev)
    pooler = _FixedPooler(t(detgen.det(tag + ".roi_feats", (P, C, 7, 7))).to(dev))
    feats = [torch.zeros(1, C, 4, 4, device=dev) for _ in range(4)]
    with torch.no_grad():
        logits, pred, obj = st(feats, bx, t(detgen.det(tag + ".prop", (1, P, C))).to(dev), pooler, None)
    np.testing.assert_allclose(pooler.rois.cpu().numpy(), GOLD[tag + ".rois"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(bx.cpu().numpy(), GOLD[tag + ".boxes_after"], rtol=1e-6, atol=1e-4)  # centres -> metres in place
    np.testing.assert_allclose(pred.cpu().numpy(), GOLD[tag + ".pred"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), GOLD[tag + ".logits"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(obj.cpu().numpy(), GOLD[tag + ".obj"], rtol=1e-4, atol=2e-4)


def _clamp_case():
    P = 48
    deltas = detgen.det("deltas.d", (P, 10), scale=0.5)
    deltas[:4, 3:6] = 12.0  # above log(100000 / 16): the scale clamp of srfdet_head.py:1580-1582
    b = mf.det_boxes("boxutil.boxes", P)
    b[..., :3] = b[..., :3] * 100.0 - 50.0
    return deltas, b[0]


def test_apply_deltas_clamp_fixture_on_hip(dev):
    """The reference's apply_deltas_lidar output WITH rows that hit the scale clamp, through srf_apply_deltas."""
    deltas, boxes = _clamp_case()
    st = heads.SingleSRFDetHeadLiDAR(**mf.STAGE_KW)
    got = ops.apply_deltas(t(deltas).to(dev), t(boxes.copy()).to(dev), st.bbox_weights[:6], st.pc_range_lidar, st.scale_clamp)
    np.testing.assert_allclose(got.cpu().numpy(), GOLD1["apply_deltas"], rtol=1e-5, atol=1e-5)


def test_stage_tail_clamp_branch_matches_reference(dev):
    """srf_stage_tail with the clamp taken: the towers are set up so that bboxes_delta reproduces the fixture's deltas
    exactly (identity-free: zero weights, the deltas as per-row bias is not expressible, so the deltas come from a one-hot
    object feature through the last Linear), and the fused kernel's boxes must equal the reference's apply_deltas output."""
    deltas, boxes = _clamp_case()
    P, C = deltas.shape[0], 128
    nn = torch.nn
    st = heads.SingleSRFDetHeadLiDAR(**mf.STAGE_KW)

    def ln_identity():
        m = nn.LayerNorm(C)
        return m.to(dev)

    # object features: row r = sqrt(C/(C-1))-scaled one-hot pattern that LayerNorm(+ReLU) maps to a vector whose only
    # positive entry is column r; with zero FFN weights norm3 leaves that pattern, the regression tower is empty, and
    # bboxes_delta.weight[:, r] * value = deltas[r]
    obj = torch.zeros(P, C)
    obj[torch.arange(P), torch.arange(P)] = 1.0
    lin1, lin2 = nn.Linear(C, 512), nn.Linear(512, C)
    for m in (lin1, lin2):
        nn.init.zeros_(m.weight)
        nn.init.zeros_(m.bias)
    norm3 = ln_identity()
    logits_fc, deltas_fc = nn.Linear(C, 10), nn.Linear(C, 10)
    with torch.no_grad():
        x = torch.nn.functional.layer_norm(obj, (C,))            # what norm3 hands to the towers (FFN contributes 0)
        # solve deltas = x @ W^T + b with b = x-independent part: x[r] = a on column r, c elsewhere
        a, c = x[0, 0].item(), x[0, 1].item()
        W = torch.zeros(10, C)
        d = t(deltas)                                           # (P, 10)
        # x @ W^T = (a - c) * W[:, r] + c * W.sum(1); choose W.sum(1) = 0 by putting -sum on an unused column
        W[:, :P] = (d / (a - c)).t()
        W[:, P] = -W[:, :P].sum(1)
        deltas_fc.weight.copy_(W)
        nn.init.zeros_(deltas_fc.bias)
    mods = [m.to(dev) for m in (lin1, lin2, logits_fc, deltas_fc)]
    lin1, lin2, logits_fc, deltas_fc = mods
    with torch.no_grad():
        want_d = torch.nn.functional.linear(torch.nn.functional.layer_norm(obj, (C,)).to(dev), deltas_fc.weight, deltas_fc.bias)
        np.testing.assert_allclose(want_d.cpu().numpy(), deltas, rtol=0, atol=2e-5)     # the construction reproduces the deltas
        _, _, pred = ops.stage_tail(obj.to(dev), (lin1, lin2), norm3, [], [], logits_fc, deltas_fc, t(boxes.copy()).to(dev),
                                    st.bbox_weights[:6], st.pc_range_lidar, st.scale_clamp)
    got, want = pred.cpu().numpy(), GOLD1["apply_deltas"]
    np.testing.assert_allclose(got[:4, 3:6], want[:4, 3:6], rtol=1e-5, atol=1e-5)       # the clamped rows: log(size) + clamp
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)


def test_ota_assigner_and_loss_on_hip_iou(dev):
    """OTAssignerSRFDet.forward + loss_ota (ota_srfdet.py:57-327, srfdet_head.py:1042-1201) with the 3-D IoU from the HIP
    rotated-IoU kernel: the IoU matrices equal the ones the reference run saw (float64 polygon clipping) to 2e-5, the
    assignments are identical, the losses agree to 1e-4 relative."""
    outs, gts, labels = ota_inputs(dev)
    from srfdet3d_amd.plugin.bbox_util import denormalize_bbox
    for h, o in ((6, outs[0]), (1, outs[1]), (2, outs[2])):
        for b in range(2):
            iou = training.bbox_overlaps_3d(denormalize_bbox(o["pred_boxes"][b], mf.NUSC_RANGE), gts[b])
            np.testing.assert_allclose(iou.cpu().numpy(), GOLD[f"ota.h{h}.iou{b}"], rtol=0, atol=2e-5)
    assigner = training.OTAssignerSRFDet(**mf2.OTA_KW)
    check_ota(assigner, outs, gts, labels)
    check_loss_ota(assigner, outs, gts, labels, rtol=1e-4)


def test_lc_training_step_two_frames_default_np(dev):
    """Config 4's per-GPU shape: srfdet_voxel_nusc_LC, TWO frames per step, the config's own np = 900, LiDAR branch frozen
    (tools/train.py:221-276): forward_train -> loss_ota -> backward -> AdamW step.  The loss arithmetic itself is pinned by
    the fixture tests above; this checks that the batched path (bs = 2: per-sample attention, the reference's cam-major
    RoI batch ids) produces finite per-layer losses whose assignment uses both samples, and gradients everywhere."""
    from srfdet3d_amd import synthetic as S, workloads
    from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
    torch.manual_seed(0)
    model = workloads.build("srfdet_voxel_nusc_LC", 900, train=True)
    training.freeze_lidar_components(model)
    model = model.to(dev).train()
    rig = [m for m in S.camera_rig(f=177.0, cx=112.0, cy=64.0)]
    pts = [t(S.nuscenes_sweep(2000 + i, 8000)).to(dev) for i in range(2)]
    img = torch.cat([t(S.camera_images(3000 + i, h=128, w=224)) for i in range(2)], 0).to(dev)
    g = torch.Generator().manual_seed(5)
    gtb, gtl = [], []
    for i in range(2):
        n = 6 + i
        b = torch.cat([torch.rand(n, 2, generator=g) * 60 - 30, torch.rand(n, 1, generator=g) * 2 - 3, torch.rand(n, 3, generator=g) * 3 + 1,
                       torch.rand(n, 1, generator=g) * 6 - 3, torch.randn(n, 2, generator=g)], 1)
        gtb.append(LiDARInstance3DBoxes(b.to(dev), box_dim=9))
        gtl.append(torch.randint(0, 10, (n,), generator=g).to(dev))
    metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=rig) for _ in range(2)]
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01)
    losses = model(return_loss=True, img=img, points=pts, img_metas=metas, gt_bboxes_3d=gtb, gt_labels_3d=gtl)
    assert set(losses) == {"loss_cls", "loss_bbox"} | {f"s.{i}.{n}" for i in range(4) for n in ("loss_cls", "loss_bbox")}
    assert all(torch.isfinite(v) and float(v) > 0 for v in losses.values())
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for k in ("bbox_head.head_series_lidar.0.output_fused_proj.weight", "bbox_head.head_series_lidar.4.bboxes_delta_lidar.weight",
              "bbox_head.img_convs.0.weight", "img_neck.fpn_convs.0.conv.weight", "bbox_head.init_proposal_boxes.weight"):
        gr = named[k].grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0, k
    assert all(p.grad is None for n, p in named.items() if n.startswith("pts_"))
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 35.0)
    opt.step()
