"""Host-side modules that are plain torch (dense backbone, FPN, VFE, BN folding, weight-layout conversion)."""
import numpy as np
import torch

import srfdet3d_amd as S
from srfdet3d_amd import sparse, workloads
from srfdet3d_amd.compat import necks
from srfdet3d_amd.plugin import backbones, norm, voxel_encoders


def test_second_and_fpn_shapes_nusc():
    m = workloads.model_cfg("srfdet_voxel_nusc_L")
    bb = S.compat.build_backbone(m.pts_backbone).eval()
    neck = S.compat.build_neck(m.pts_neck).eval()
    x = torch.randn(1, 256, 48, 40)
    with torch.no_grad():
        o = bb(x)
        f = neck(o)
    assert [tuple(t.shape) for t in o] == [(1, 128, 48, 40), (1, 256, 24, 20)]
    assert [tuple(t.shape) for t in f] == [(1, 128, 48, 40), (1, 128, 24, 20), (1, 128, 12, 10), (1, 128, 6, 5)]
    assert sum(isinstance(l, torch.nn.Conv2d) for l in bb.modules()) == 12  # SURVEY.md finding 5: 12 dense 3x3 convs


def test_fpn_without_extra_convs_uses_maxpool_kitti():
    m = workloads.model_cfg("srfdet_voxel_kitti_L")
    neck = S.compat.build_neck(m.pts_neck).eval()
    assert len(neck.fpn_convs) == 2
    with torch.no_grad():
        f = neck((torch.randn(1, 128, 40, 36), torch.randn(1, 256, 20, 18)))
    assert [tuple(t.shape) for t in f] == [(1, 256, 40, 36), (1, 256, 20, 18), (1, 256, 10, 9), (1, 256, 5, 5)]
    torch.testing.assert_close(f[2], f[1][:, :, ::2, ::2])


def test_hard_simple_vfe_mean():
    vfe = voxel_encoders.HardSimpleVFE(5)
    v = torch.zeros(3, 10, 5)
    v[0, :2] = torch.tensor([[1., 2, 3, 4, 5], [3., 4, 5, 6, 7]])
    v[1, :1] = 1.0
    v[2, :10] = torch.arange(50.).view(10, 5)
    out = vfe(v, torch.tensor([2, 1, 10]), None)
    torch.testing.assert_close(out[0], torch.tensor([2., 3, 4, 5, 6]))
    torch.testing.assert_close(out[2], v[2].mean(0))


def test_bn_fold_matches_eval_batchnorm():
    bn = torch.nn.BatchNorm1d(32, eps=1e-3).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    a, b = sparse._fold_bn(bn)
    x = torch.randn(100, 32)
    torch.testing.assert_close(torch.addcmul(b, x, a), bn(x), rtol=1e-6, atol=1e-6)
    a2, _ = sparse._fold_bn(bn)
    assert a2.data_ptr() == a.data_ptr()  # cached
    with torch.no_grad():
        bn.running_var.mul_(2.0)
    a3, _ = sparse._fold_bn(bn)
    assert not torch.equal(a3, a)  # cache invalidated by the in-place update


def test_spconv2_weight_layout_is_converted_on_load():
    conv = sparse.SubMConv3d(4, 8, 3, padding=1, bias=False)
    w1 = torch.randn(3, 3, 3, 4, 8)
    conv.load_state_dict({"weight": w1.permute(4, 0, 1, 2, 3).contiguous()})  # spconv 2.x: (Cout, kD, kH, kW, Cin)
    torch.testing.assert_close(conv.weight.data, w1)
    conv.load_state_dict({"weight": w1 * 2})
    torch.testing.assert_close(conv.weight.data, w1 * 2)


def test_sparse_encoder_layer_inventory_nusc():
    enc = S.compat.build_middle_encoder(workloads.model_cfg("srfdet_voxel_nusc_L").pts_middle_encoder)
    convs = [m for m in enc.modules() if isinstance(m, sparse._SparseConv)]
    assert sum(c.subm for c in convs) == 17 and sum(not c.subm for c in convs) == 4
    assert enc.conv_out[0].kernel_size == [3, 1, 1] and enc.conv_out[0].stride == [2, 1, 1]
    l3 = enc.encoder_layers.encoder_layer3[2][0]
    assert l3.padding == [0, 1, 1] and l3.out_channels == 128
    enc_k = S.compat.build_middle_encoder(workloads.model_cfg("srfdet_voxel_kitti_L").pts_middle_encoder)
    convs = [m for m in enc_k.modules() if isinstance(m, sparse._SparseConv)]
    assert [c.out_channels for c in convs] == [16, 16, 32, 32, 32, 64, 64, 64, 64, 64, 64, 128]


def test_naive_sync_bn_single_process_is_plain_bn():
    bn = norm.NaiveSyncBatchNorm1dCustom(6, eps=1e-3, momentum=0.01)
    ref = torch.nn.BatchNorm1d(6, eps=1e-3, momentum=0.01)
    ref.load_state_dict(bn.state_dict())
    x = torch.randn(50, 6)
    torch.testing.assert_close(bn(x), ref(x))
    bn.eval(), ref.eval()
    torch.testing.assert_close(bn(x), ref(x))


def test_hungarian_assigner_one_to_one():
    """HungarianAssignerSRFDet (hungarian_assigner_srfdet.py:14-129): each ground truth gets exactly one prediction, the
    cheapest overall; everything else is background; empty ground truth -> all background."""
    from srfdet3d_amd.compat.registry import BBOX_ASSIGNERS
    from srfdet3d_amd.plugin.bbox_util import normalize_bbox
    import srfdet3d_amd.plugin.training  # noqa: F401  (registers the assigner)
    asg = BBOX_ASSIGNERS.build(dict(type="HungarianAssignerSRFDet", cls_cost=dict(type="FocalLossCost", weight=2.0),
                                    reg_cost=dict(type="BBox3DL1Cost", weight=0.25), pc_range=[-50, -50, -5, 50, 50, 3]))
    g = torch.Generator().manual_seed(0)
    gt = torch.cat([torch.rand(4, 3, generator=g) * 40 - 20, torch.rand(4, 3, generator=g) * 3 + 1,
                    torch.rand(4, 1, generator=g) * 3 - 1.5, torch.zeros(4, 2)], 1)
    gt_labels = torch.tensor([1, 3, 0, 2])
    pred = torch.randn(12, 10, generator=g)
    logits = torch.randn(12, 5, generator=g)
    want_rows = [7, 2, 9, 4]
    pred[want_rows] = normalize_bbox(gt) + 0.01 * torch.randn(4, 10, generator=g)
    logits[want_rows, gt_labels] += 6.0
    res = asg.assign(pred, logits, gt, gt_labels)
    assert res.num_gts == 4 and (res.gt_inds > 0).sum() == 4
    assert res.gt_inds[want_rows].tolist() == [1, 2, 3, 4]
    assert res.labels[want_rows].tolist() == gt_labels.tolist()
    assert (res.labels[res.gt_inds == 0] == -1).all()
    empty = asg.assign(pred, logits, gt[:0], gt_labels[:0])
    assert (empty.gt_inds == 0).all() and empty.num_gts == 0
    default = BBOX_ASSIGNERS.build(dict(type="HungarianAssignerSRFDet"))     # mmdet's default cost names resolve too
    assert (default.assign(pred, logits, gt, gt_labels).gt_inds > 0).sum() == 4


def test_results_from_static_host_unpacking():
    """Host side of the fixed-shape NMS selection (heads.results_from_static, numpy on the read-back rows): survivors, the
    max_per_img cut in descending score, the post_center_range filter, and None on a capacity overflow -- against the plain torch
    formulation of srfdet_head.py:1288-1310."""
    import numpy as np
    import torch
    from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
    from srfdet3d_amd.plugin import heads
    hd = object.__new__(heads.SRFDetHead)
    torch.nn.Module.__init__(hd)
    hd.test_cfg = dict(max_per_img=5, post_center_range=[-10.0, -10.0, -5.0, 10.0, 10.0, 5.0])
    g = torch.Generator().manual_seed(0)
    L, D = 16, 9
    packed = torch.zeros(2, L, D + 2)
    packed[..., :3] = torch.rand(2, L, 3, generator=g) * 30 - 15
    packed[..., 3:D] = torch.rand(2, L, D - 3, generator=g)
    packed[..., D] = torch.rand(2, L, generator=g)
    packed[..., D + 1] = torch.randint(0, 10, (2, L), generator=g).float()
    counts = torch.tensor([[9, 12], [3, 3]], dtype=torch.int32)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)] * 2
    res = hd.results_from_static(packed, counts, metas)
    for i in range(2):
        rows = packed[i, :int(counts[i, 0])]
        boxes, scores, labels = rows[:, :D], rows[:, D], rows[:, D + 1].long()
        if len(rows) > 5:
            top = scores.sort(descending=True)[1][:5]
            boxes, scores, labels = boxes[top], scores[top], labels[top]
        rng = torch.tensor(hd.test_cfg["post_center_range"])
        keep = (boxes[:, :3] >= rng[:3]).all(1) & (boxes[:, :3] <= rng[3:]).all(1)
        assert torch.equal(res[i][0].tensor, boxes[keep]) and torch.equal(res[i][1], scores[keep]) and torch.equal(res[i][2], labels[keep])
        assert res[i][2].dtype == torch.int64
    # the rows are copies: the read-back buffer may be overwritten by the next frame
    before = res[0][1].clone()
    packed.zero_()
    assert torch.equal(res[0][1], before)
    assert hd.results_from_static(packed, torch.tensor([[3, L + 1], [0, 0]], dtype=torch.int32), metas) is None
