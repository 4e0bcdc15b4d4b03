"""LiDAR+camera pieces against fixtures produced by the reference's own Python (tests/golden/fusion_nusc.npz):
VoVNet-99 and the 5-stage fusion decoder (image DPG, img_convs, per-camera RoIs, camera sum, fused projection)."""
import os

import numpy as np
import pytest
import torch

import detgen
from oracle import pipeline
from srfdet3d_amd import synthetic as S, workloads
from srfdet3d_amd.compat.registry import build_head
from srfdet3d_amd.plugin.vovnet import VoVNet

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "fusion_nusc.npz"))
Pc, N_CAM = 16, 6


def test_vovnet99_matches_reference():
    net = VoVNet("V-99-eSE", input_ch=3, out_features=["stage2", "stage3", "stage4", "stage5"]).eval()
    detgen.load_det_params(net, "vov.")
    with torch.no_grad():
        out = net(torch.from_numpy(detgen.det("vov.img", (1, 3, 32, 48))))
    for k, v in out.items():
        ref = GOLD["vov." + k]
        assert tuple(v.shape) == ref.shape
        np.testing.assert_allclose(v.numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def lc_head():
    m = workloads.model_cfg("srfdet_voxel_nusc_LC")
    hc = dict(m.bbox_head)
    hc.update(num_proposals=Pc, train_cfg=None, test_cfg=m.test_cfg, use_img=True)
    hd = build_head(hc).eval()
    detgen.load_det_params(hd, "headlc.")
    pf = [torch.from_numpy(detgen.det(f"headlc.feat{i}", (1, 128, s, s), scale=0.5)) for i, s in enumerate((184, 92, 46, 23))]
    imf = [torch.from_numpy(detgen.det(f"headlc.img{i}", (1, N_CAM, 256, h, w), scale=0.5))
           for i, (h, w) in enumerate(((32, 56), (16, 28), (8, 14), (4, 7)))]
    metas = [dict(lidar2img=[m for m in S.camera_rig(f=177.0, cx=112.0, cy=64.0)])]
    return hd, pf, imf, metas


def test_cpu_fusion_head_stage_by_stage_matches_reference():
    hd, pf, imf, metas = lc_head()
    forced = list(zip(GOLD["headlc.stage_in_boxes"], GOLD["headlc.stage_in_prop"]))
    logits, boxes = pipeline.head_forward(hd, imf, pf, metas, stage_inputs=forced)
    np.testing.assert_allclose(boxes.numpy(), GOLD["headlc.boxes"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logits.numpy(), GOLD["headlc.logits"], rtol=1e-4, atol=2e-4)


@pytest.mark.gpu
def test_gpu_fusion_head_stage_by_stage_matches_reference(dev):
    """HIP box->BEV+6-camera RoIs, HIP gathers on both pyramids, camera sum, fused projection, stage arithmetic."""
    hd, pf, imf, metas = lc_head()
    hd = hd.to(dev)
    pf = [f.to(dev).contiguous(memory_format=torch.channels_last) for f in pf]
    with torch.no_grad():
        imf = [hd.img_convs[i](f.to(dev).flatten(0, 1)).unflatten(0, (1, N_CAM)) for i, f in enumerate(imf)]
    lo = torch.tensor(hd.pc_range[:3], device=dev)
    ext = torch.tensor(hd.pc_range[3:], device=dev) - lo
    for s, stage in enumerate(hd.head_series_lidar):
        bx = torch.from_numpy(GOLD["headlc.stage_in_boxes"][s].copy()).to(dev)
        prop = torch.from_numpy(GOLD["headlc.stage_in_prop"][s].copy()).to(dev)
        with torch.no_grad():
            logits, pred, _ = stage(imf, pf, bx, prop, hd.roi_extractor_lidar, metas, pooler_img=hd.roi_extractor_img)
        pred = pred.clone()
        pred[..., :3] = pred[..., :3] * ext + lo
        np.testing.assert_allclose(pred.cpu().numpy(), GOLD["headlc.boxes"][s], rtol=0, atol=1e-4)
        np.testing.assert_allclose(logits.cpu().numpy(), GOLD["headlc.logits"][s], rtol=1e-4, atol=3e-4)
    # free-running loop through SRFDetHead.forward (image DPG + img_convs on MIOpen included).  The stage contract
    # (1e-4) is the teacher-forced loop above; here the img_convs' different summation order feeds a random-weight
    # stage that amplifies it, so the first stage is only required to stay within 1e-3
    hd2, pf2, imf2, metas2 = lc_head()
    hd2 = hd2.to(dev)
    with torch.no_grad():
        lg, bx = hd2([f.to(dev) for f in imf2], [f.to(dev) for f in pf2], metas2)
    np.testing.assert_allclose(bx.cpu().numpy()[0], GOLD["headlc.boxes"][0], rtol=0, atol=1e-3)
