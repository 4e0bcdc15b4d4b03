"""The LC frame at the headline size (30k points, six 928 x 1600 views, np = 200) against an implementation that shares no
convolution code with it: the camera branch (VoVNet-99 -> image FPN -> `img_convs`; vovnet.py:116-375, srfdet.py:175-202,
srfdet_head.py:404-416) run by torch on the CPU, i.e. DIRECT convolutions, against the HIP executor whose 3x3 layers are
Winograd F(4x4, 3x3) / F(2x2, 3x3).

VERDICT r3, missing 3: north_star's contract is "fp32 box params within 1e-4 on identical inputs"; until now the camera
features were held to 2e-4 of each level's maximum and the decoder stages to 1e-4 GIVEN identical features, but no test fed
Winograd-produced features through the fusion stages (srfdet_head.py:2255-2329, :2424-2566) and compared boxes."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import pipeline
from srfdet3d_amd import nhwc, synthetic as S, workloads
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes

pytestmark = pytest.mark.gpu
NP = 200


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


@pytest.fixture(scope="module")
def lc_full():
    """The np = 200 LC model and its camera pyramid by torch-CPU direct convolutions (the slow part: computed once)."""
    torch.manual_seed(0)
    cpu = workloads.build("srfdet_voxel_nusc_LC", NP).eval()
    _randomize_bn(cpu, 0)
    img = torch.from_numpy(S.camera_images(3000))                      # (1, 6, 3, 928, 1600)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in S.camera_rig()])]
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        raw = cpu.extract_img_feat(img, copy.deepcopy(metas))          # FPN outputs, before img_convs
    return dict(cpu=cpu, img=img, metas=metas, raw=raw)


def test_camera_branch_at_full_size_matches_torch_cpu(lc_full, dev):
    """Every level of the camera pyramid after `img_convs` within 2e-4 of its maximum (srf_stem_conv_nchw, srf_wino43 on every
    3x3 layer from stage 2 on, srf_wino3x3, srf_conv_gemm_nhwc, srf_conv1x1_nhwc_direct / _pooled / _topdown, the streaming
    kernels -- the kernels that are 90 % of the headline frame, at the size the headline is quoted on)."""
    cpu, img, metas = lc_full["cpu"], lc_full["img"], lc_full["metas"]
    gpu = copy.deepcopy(cpu).to(dev)
    with torch.no_grad():
        got = gpu.extract_img_feat(img.to(dev), copy.deepcopy(metas))
        got = [g.float().cpu() for g in gpu.bbox_head._img_convs_only(got)]
        want = cpu.bbox_head._img_convs_only(lc_full["raw"])
    assert len(got) == len(want) == 4
    for lvl, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape == (1, 6, 128, 232 >> lvl, 400 >> lvl)
        err = (a - b).abs().max().item()
        assert err <= 2e-4 * b.abs().max().item(), (lvl, err, b.abs().max().item())
        assert b.abs().max().item() > 1e-3


def test_lc_boxes_from_winograd_features_match_direct_convolution_features(lc_full, dev):
    """Box parameters of every decoder stage, HIP frame (camera features on the Winograd kernels, fusion RoI gather, stage
    kernels) against the oracle pipeline fed the torch-CPU DIRECT-convolution camera features: <= 1e-4 per stage, stages
    teacher-forced (each oracle stage starts from the boxes / proposal features the HIP stage started from, so the comparison
    isolates one stage's arithmetic on the two feature sets instead of the free-running loop's amplification of rounding --
    DESIGN.md section 2).  The LiDAR pyramid is shared (its parity is bit-exact elsewhere); what differs between the two sides
    is exactly the camera branch's arithmetic and the decoder's."""
    cpu, img, metas = lc_full["cpu"], lc_full["img"], lc_full["metas"]
    gpu = copy.deepcopy(cpu).to(dev)
    pts = torch.from_numpy(S.nuscenes_sweep(2000, 30000)).to(dev)
    rec = []
    hooks = [st.register_forward_pre_hook(lambda m, a: rec.append((a[2].detach().clone().cpu().numpy(),
                                                                  a[3].detach().clone().cpu().numpy().reshape(1, NP, -1))))
             for st in gpu.bbox_head.head_series_lidar]
    with torch.no_grad():
        mt = copy.deepcopy(metas)
        img_feats, pt_feats = gpu.extract_feat(img.to(dev), [pts], mt)
        logits, boxes = gpu.bbox_head(img_feats, pt_feats, mt)
    for h in hooks:
        h.remove()
    assert boxes.shape == (5, 1, NP, 10) and len(rec) == 5
    ref_logits, ref_boxes = pipeline.head_forward(cpu.bbox_head, lc_full["raw"], [f.cpu() for f in pt_feats], copy.deepcopy(metas),
                                                  stage_inputs=rec)
    got, want = boxes.cpu().numpy(), ref_boxes.numpy()
    per_stage = np.abs(got - want).reshape(5, -1).max(1)
    print("LC box-parameter error per stage (Winograd camera features vs direct-convolution features):", per_stage)
    assert per_stage.max() <= 1e-4, per_stage
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits.numpy(), rtol=1e-4, atol=2e-4)
    # how much of that is the camera branch: the same oracle stages on the HIP (Winograd) FPN outputs (`img_convs` then run
    # as CPU direct convolutions on both sides, so the last Winograd layer counts as "decoder side" in this split)
    assert not isinstance(img_feats, nhwc.ConsumedLevels)
    hip_raw = [f.float().cpu() for f in img_feats]
    _, ref2 = pipeline.head_forward(cpu.bbox_head, hip_raw, [f.cpu() for f in pt_feats], copy.deepcopy(metas), stage_inputs=rec)
    decoder_side = np.abs(got - ref2.numpy()).reshape(5, -1).max(1)
    camera_side = np.abs(ref2.numpy() - want).reshape(5, -1).max(1)
    print("  HIP stages + img_convs against oracle stages on the SAME (HIP) FPN outputs:", decoder_side)
    print("  oracle stages on HIP FPN outputs against oracle stages on torch-CPU FPN outputs:", camera_side)
    assert camera_side.max() <= 1e-4
