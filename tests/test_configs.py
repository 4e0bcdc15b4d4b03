"""The drop-in boundary on the host: the reference's config files load unchanged, every registry name resolves,
and the built models carry the reference's state_dict keys."""
import json
import os

import pytest
import torch

import srfdet3d_amd as S
from srfdet3d_amd import workloads
from srfdet3d_amd.compat.config import Config

REF = "/root/reference"
NAMES = {"srfdet_voxel_nusc_L": "configs/nus/srfdet_voxel_nusc_L.py", "srfdet_voxel_nusc_LC": "configs/nus/srfdet_voxel_nusc_LC.py",
         "srfdet_voxel_kitti_L": "configs/kitti/srfdet_voxel_kitti_L.py", "srfdet_dvoxel_waymo_L": "configs/waymo/srfdet_dvoxel_waymo_L.py",
         "srfdet_pillar_nusc_L": "configs/nus/srfdet_pillar_nusc_L.py", "srfdet_voxel_kitti_LC": "configs/kitti/srfdet_voxel_kitti_LC.py",
         "srfdet_pillar_v299_nusc_LC": "configs/nus/srfdet_pillar_v299_nusc_LC.py",
         "srfdet_pillar_r50_nusc_LC": "configs/nus/srfdet_pillar_r50_nusc_LC.py",
         "srfdet_voxel_r50_nusc_LC": "configs/nus/srfdet_voxel_r50_nusc_LC.py", "srfdet_dvoxel_nusc_L": "configs/others/srfdet_dvoxel_nusc_L.py",
         "srfdet_dvoxel_waymo_LC": "configs/others/srfdet_dvoxel_waymo_LC.py"}


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", sorted(NAMES))
def test_workload_json_equals_live_reference_config(name):
    cfg = Config.fromfile(os.path.join(REF, NAMES[name]))
    assert cfg.plugin is True and cfg.plugin_dir == "mmdet3d_plugin"
    assert _plain(cfg.model) == _plain(workloads.model_cfg(name))


def test_cfg_options_override():
    m = workloads.model_cfg("srfdet_voxel_nusc_L", **{"bbox_head.num_proposals": 200})
    assert m.bbox_head.num_proposals == 200 and m["pts_voxel_layer"]["max_voxels"] == (120000, 160000)
    c = Config(a=dict(b=[dict(c=1), dict(c=2)]))
    c.merge_from_dict({"a.b.1.c": 5, "a.d.e": "x"})
    assert c.a.b[1]["c"] == 5 and c.a.d.e == "x"


@pytest.mark.parametrize("name,np_,n_params_m", [("srfdet_voxel_nusc_L", 900, None), ("srfdet_voxel_kitti_L", 100, None),
                                                ("srfdet_dvoxel_waymo_L", 900, None)])
def test_models_build_with_reference_key_names(name, np_, n_params_m):
    model = workloads.build(name, np_)
    keys = set(model.state_dict().keys())
    must = ["pts_middle_encoder.conv_input.0.weight", "pts_middle_encoder.conv_out.0.weight", "pts_backbone.blocks.0.0.weight",
            "pts_neck.lateral_convs.0.conv.weight", "pts_neck.fpn_convs.0.bn.running_mean",
            "bbox_head.init_proposal_boxes.weight", "bbox_head.dpg_dw_convs_lidar.0.conv.weight", "bbox_head.dpg_fc1_lidar.weight",
            "bbox_head.head_series_lidar.4.self_attn_lidar.in_proj_weight",
            "bbox_head.head_series_lidar.0.inst_interact_lidar.dynamic_layer.weight",
            "bbox_head.head_series_lidar.0.bboxes_delta_lidar.bias", "bbox_head.code_weights"]
    if "nusc" in name:
        must += ["pts_middle_encoder.encoder_layers.encoder_layer1.0.conv1.weight",
                 "pts_middle_encoder.encoder_layers.encoder_layer1.2.0.weight", "pts_neck.fpn_convs.3.conv.weight"]
    else:
        must += ["pts_voxel_encoder.vfe_layers.0.linear.weight", "pts_voxel_encoder.vfe_layers.0.norm.running_var",
                 "pts_voxel_encoder.cen2point_pos_enc.0.weight"]
    for k in must:
        assert k in keys, k
    sd = model.state_dict()
    if "nusc" in name:
        assert tuple(sd["pts_middle_encoder.conv_input.0.weight"].shape) == (3, 3, 3, 5, 16)
        assert tuple(sd["bbox_head.dpg_fc1_lidar.weight"].shape) == (1024, 23 * 23)
        assert tuple(sd["bbox_head.init_proposal_boxes.weight"].shape) == (4 * 900, 10)
        stage = sum(p.numel() for p in model.bbox_head.head_series_lidar[0].parameters())
        assert stage == 2144596  # parameter count of one reference LiDAR stage at the nusc_L settings
    if "kitti" in name:
        assert tuple(sd["bbox_head.dpg_fc1_lidar.weight"].shape) == (1024, 25 * 22)
        assert sum(p.numel() for p in model.bbox_head.head_series_lidar[0].parameters()) == 12757387


def test_lc_head_builds_with_fusion_stage():
    m = workloads.model_cfg("srfdet_voxel_nusc_LC")
    hc = dict(m.bbox_head)
    hc.update(num_proposals=16, train_cfg=None, test_cfg=m.test_cfg, use_img=True)
    head = S.compat.build_head(hc)
    assert sum(p.numel() for p in head.head_series_lidar[0].parameters()) == 2177492
    keys = set(head.state_dict())
    assert {"img_convs.3.weight", "dpg_fc2_img.weight", "head_series_lidar.0.output_fused_proj.weight"} <= keys


@pytest.mark.parametrize("name", ["srfdet_voxel_r50_nusc_LC", "srfdet_dvoxel_nusc_L"])
def test_additional_configs_build(name):
    """every config of the reference except the DCNv2 one builds from its unchanged model dict (the large VoVNet ones are
    built in the GPU tests)."""
    m = workloads.build(name, 16)
    assert sum(p.numel() for p in m.parameters()) > 1e6


def test_dcn_config_is_refused_loudly():
    with pytest.raises(NotImplementedError, match="deformable"):
        workloads.build("srfdet_dvoxel_waymo_LC", 16)


def test_resnet_state_dict_names_follow_mmdet():
    from srfdet3d_amd.compat.registry import build_backbone
    import torch
    r = build_backbone(dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                            norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, style="pytorch")).eval()
    keys = set(r.state_dict())
    for k in ("conv1.weight", "bn1.running_mean", "layer1.0.downsample.0.weight", "layer1.0.downsample.1.weight",
              "layer3.5.conv2.weight", "layer4.2.bn3.bias"):
        assert k in keys
    assert r.layer2[0].conv2.stride == (2, 2) and r.layer2[0].conv1.stride == (1, 1)      # pytorch style
    assert not r.conv1.weight.requires_grad and not r.layer1[0].conv1.weight.requires_grad and r.layer2[0].conv1.weight.requires_grad
    with torch.no_grad():
        outs = r(torch.zeros(1, 3, 64, 96))
    assert [o.shape[1] for o in outs] == [256, 512, 1024, 2048] and outs[0].shape[-2:] == (16, 24)
