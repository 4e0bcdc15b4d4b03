"""Host-side arithmetic against the round-2 fixtures produced by the REFERENCE's own Python
(tests/golden/make_fixtures_r2.py -> tests/golden/extra_r2.npz): SECONDCustom, the stage at the KITTI arguments and at
P = 200 / 900, the OTA assigner and loss_ota.  CPU torch only -- the HIP paths meet the same fixtures in
tests/test_gpu_fixtures_r2.py.  Inputs and weights are regenerated from (name, shape) on both sides, so a pass also proves
parameter names and shapes equal the reference's."""
import os

import numpy as np
import pytest
import torch

import detgen
import make_fixtures as mf
import make_fixtures_r2 as mf2
from srfdet3d_amd.plugin import backbones, heads, training

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "extra_r2.npz"))
t = torch.from_numpy


def test_second_custom_matches_reference():
    net = backbones.SECONDCustom(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2]).eval()
    detgen.load_det_params(net, "second.")
    with torch.no_grad():
        o = net(t(detgen.det("second.x", (1, 256, 24, 20), scale=0.5)))
    np.testing.assert_allclose(o[0].numpy(), GOLD["second.out0"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o[1].numpy(), GOLD["second.out1"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag,kw,P,C", [("kstage", mf2.KSTAGE_KW, 100, 256), ("lstage200", mf.STAGE_KW, 200, 128),
                                        ("lstage900", mf.STAGE_KW, 900, 128)])
def test_stage_arithmetic_matches_reference(tag, kw, P, C):
    st = heads.SingleSRFDetHeadLiDAR(**kw).eval()
    detgen.load_det_params(st, tag + ".")
    roi = t(detgen.det(tag + ".roi_feats", (P, C, 7, 7))).flatten(2).permute(0, 2, 1).contiguous()
    with torch.no_grad():
        logits, pred, obj = st._refine(roi, t(GOLD[tag + ".boxes_after"].copy()), t(detgen.det(tag + ".prop", (1, P, C))), 1, P)
    np.testing.assert_allclose(pred.numpy(), GOLD[tag + ".pred"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(logits.numpy(), GOLD[tag + ".logits"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(obj.numpy(), GOLD[tag + ".obj"], rtol=1e-4, atol=2e-4)


def ota_inputs(dev="cpu"):
    stages = [mf2.ota_case(f"ota.s{i}", 64, [7, 5]) for i in range(3)]
    gts = [t(g).to(dev) for g in stages[0][2]]
    labels = [t(l).to(dev) for l in stages[0][3]]
    outs = [dict(pred_logits=t(s[0]).to(dev), pred_boxes=t(s[1]).to(dev)) for s in stages]
    return outs, gts, labels


def check_ota(assigner, outs, gts, labels):
    for head_idx, o in ((6, outs[0]), (1, outs[1]), (2, outs[2])):
        res = assigner(o, gts, labels, head_idx)
        for b, (fg, gi) in enumerate(res):
            np.testing.assert_array_equal(fg.cpu().numpy(), GOLD[f"ota.h{head_idx}.fg{b}"])
            np.testing.assert_array_equal(gi.cpu().numpy(), GOLD[f"ota.h{head_idx}.gt{b}"])
    fg, gi = assigner.single_assigner(outs[0]["pred_boxes"][0], outs[0]["pred_logits"][0], gts[0][:0], labels[0][:0], 6)
    np.testing.assert_array_equal(fg.cpu().numpy(), GOLD["ota.empty.fg"])
    assert gi.numel() == 0


def check_loss_ota(assigner, outs, gts, labels, rtol):
    hd = object.__new__(heads.SRFDetHead)
    torch.nn.Module.__init__(hd)
    hd.assigner, hd.num_heads, hd.deep_supervision, hd.num_classes, hd.sync_cls_avg_factor = assigner, 6, True, 10, True
    hd.pc_range = mf.NUSC_RANGE
    hd.code_weights = torch.nn.Parameter(torch.tensor([1.0] * 8 + [0.2, 0.2], device=gts[0].device), requires_grad=False)
    hd.loss_cls = training.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="sum", loss_weight=2.0)
    hd.loss_bbox = training.L1Loss(reduction="sum", loss_weight=0.25)
    outputs = dict(outs[0], aux_outputs=outs[1:])
    losses = hd.loss_ota(outputs, [mf2._GtBoxes(g) for g in gts], labels)
    assert set(losses) == {"loss_cls", "loss_bbox", "s.0.loss_cls", "s.0.loss_bbox", "s.1.loss_cls", "s.1.loss_bbox"}
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(GOLD["ota.loss." + k]), rtol=rtol)


def test_ota_assigner_and_losses_match_reference_given_its_ious(monkeypatch):
    """Assignment, classification and box losses against the reference with the 3-D IoU (the third-party part) replaced by
    the matrices the reference run saw: everything else -- in-box / in-centre tests, costs, dynamic k, tie handling, the
    loss normalisation -- is this repo's code on the CPU."""
    outs, gts, labels = ota_inputs()
    calls = {"n": 0}
    order = [(6, 0), (6, 1), (1, 0), (1, 1), (2, 0), (2, 1)]

    def fake_iou(a, b):
        if b.shape[0] == 0:
            return a.new_zeros((a.shape[0], 0))
        h, s = order[calls["n"] % 6]
        calls["n"] += 1
        return t(GOLD[f"ota.h{h}.iou{s}"].copy())

    monkeypatch.setattr(training, "bbox_overlaps_3d", fake_iou)
    assigner = training.OTAssignerSRFDet(**mf2.OTA_KW)
    check_ota(assigner, outs, gts, labels)
    calls["n"] = 0
    order[:] = [(6, 0), (6, 1), (1, 0), (1, 1), (2, 0), (2, 1)]
    check_loss_ota(assigner, outs, gts, labels, rtol=1e-5)
