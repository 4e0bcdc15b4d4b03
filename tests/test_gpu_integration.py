"""Integration parity: the whole hot path of srfdet_voxel_nusc_L (np=200, seeded random weights, randomised BN
statistics) on one synthetic 30k-point sweep, HIP path vs the CPU oracle pipeline (oracle/pipeline.py)."""
import numpy as np
import pytest
import torch

from oracle import pipeline
from srfdet3d_amd import synthetic as S, workloads
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes

pytestmark = pytest.mark.gpu


def _randomize_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


@pytest.fixture(scope="module")
def setup(dev):
    torch.manual_seed(0)
    cpu = workloads.build("srfdet_voxel_nusc_L", 200).eval()
    _randomize_bn(cpu)
    import copy
    gpu = copy.deepcopy(cpu).to(dev)
    pts = S.nuscenes_sweep(2000)
    return cpu, gpu, pts


def test_sparse_stage_is_exact_and_dense_stage_close(setup, dev):
    cpu, gpu, pts = setup
    vf, coors = pipeline.voxel_features(cpu, [pts])
    bev_ref = pipeline.sparse_encoder(cpu.pts_middle_encoder, vf, coors, 1)
    with torch.no_grad():
        voxels, num, gc = gpu.voxelize([torch.from_numpy(pts).to(dev)])
        gvf = gpu.pts_voxel_encoder(voxels, num, gc)
        np.testing.assert_array_equal(gc.cpu().numpy(), coors)          # bit-exact voxel indices, first-seen order
        np.testing.assert_array_equal(gvf.cpu().numpy(), vf)            # mean in slot order: exact
        bev = gpu.pts_middle_encoder(gvf, gc, 1)
        assert bev.shape == (1, 256, 184, 184)
        # 21 sparse convs + BN + ReLU + residuals: identical f32 fma chains on both sides
        assert np.array_equal(bev.cpu().numpy(), bev_ref), f"max diff {np.abs(bev.cpu().numpy() - bev_ref).max()}"
        feats = gpu.pts_neck(gpu.pts_backbone(bev))
        ref = cpu.pts_neck(cpu.pts_backbone(torch.from_numpy(bev_ref)))
    for f, r in zip(feats, ref):
        scale = r.abs().max().item()
        assert (f.cpu() - r).abs().max().item() <= 2e-4 * scale  # MIOpen vs CPU conv: different summation order


def test_decoder_stage_by_stage_on_real_features(setup, dev):
    """GPU decoder vs the CPU oracle decoder on the SAME (GPU-produced) pyramid, every stage fed the GPU stage's inputs."""
    cpu, gpu, pts = setup
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    rec = []
    hooks = [st.register_forward_pre_hook(lambda m, a: rec.append((a[1].detach().clone().cpu().numpy(),
                                                                  a[2].detach().clone().cpu().numpy().reshape(1, 200, -1))))
             for st in gpu.bbox_head.head_series_lidar]
    with torch.no_grad():
        feats = gpu.extract_point_features([torch.from_numpy(pts).to(dev)])
        logits, boxes = gpu.bbox_head(None, feats, metas)
    for h in hooks:
        h.remove()
    ref_logits, ref_boxes = pipeline.head_forward(cpu.bbox_head, None, [f.cpu() for f in feats], metas, stage_inputs=rec)
    np.testing.assert_allclose(boxes.cpu().numpy(), ref_boxes.numpy(), rtol=0, atol=1e-4)   # north-star tolerance
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits.numpy(), rtol=1e-4, atol=2e-4)
    # decode (pre-NMS tensors, the parity contract of SURVEY.md a18)
    s, b = gpu.bbox_head.decode(logits, boxes)
    rs, rb = cpu.bbox_head.decode(ref_logits, ref_boxes)
    np.testing.assert_allclose(b.cpu().numpy(), rb.numpy(), rtol=1e-5, atol=2e-4)
    res = gpu.simple_test(None, [torch.from_numpy(pts).to(dev)], metas)
    assert set(res[0]["pts_bbox"].keys()) == {"boxes_3d", "scores_3d", "labels_3d"}


def test_hip_graph_tail_equals_eager(setup, dev):
    cpu, gpu, pts = setup
    import copy
    gpu = copy.deepcopy(gpu)
    gpu.bbox_head.test_cfg = dict(gpu.bbox_head.test_cfg, score_thr=0.02)   # random weights: make sure there ARE detections
    g = copy.deepcopy(gpu).enable_hip_graphs(whole_frame=False)   # tail-only graph: what the dynamic-voxel configs use
    assert g._graphed_frame is None
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    for seed in (2000, 2001, 2000, 2002, 2001):
        p = torch.from_numpy(S.nuscenes_sweep(seed)).to(dev)
        with torch.no_grad():
            a = gpu.simple_test(None, [p], metas)[0]["pts_bbox"]
            b = g.simple_test(None, [p], metas)[0]["pts_bbox"]
        assert a["scores_3d"].numel() > 5, "the comparison needs detections"
        # same kernels, same order; MIOpen may pick a different conv solver between the eager and the captured run,
        # so equality is asserted to float tolerance rather than bitwise
        assert torch.equal(a["labels_3d"], b["labels_3d"])
        torch.testing.assert_close(a["scores_3d"], b["scores_3d"], rtol=0, atol=1e-5)
        torch.testing.assert_close(a["boxes_3d"].tensor, b["boxes_3d"].tensor, rtol=0, atol=1e-4)


@pytest.mark.parametrize("name,sweep,npts,np_", [("srfdet_voxel_kitti_L", "kitti_sweep", 17000, 100),
                                                 ("srfdet_dvoxel_waymo_L", "waymo_sweep", 60000, 64)])
def test_dynamic_voxel_configs_against_oracle(name, sweep, npts, np_, dev):
    """KITTI (config C1) and Waymo (C5) paths: dynamic voxelization -> DynamicVFECustom -> sparse encoder, HIP vs the CPU
    oracle pipeline.  Voxel coordinates (sorted unique) must be bit-exact; voxel features go through a Linear+BN on
    rocBLAS vs CPU BLAS, so they and everything downstream are compared with a float tolerance."""
    import copy
    torch.manual_seed(1)
    cpu = workloads.build(name, np_).eval()
    _randomize_bn(cpu, 1)
    gpu = copy.deepcopy(cpu).to(dev)
    seed = 1000 if "kitti" in name else 5000
    pts = getattr(S, sweep)(seed, npts)
    vf, vc = pipeline.voxel_features(cpu, [pts])
    with torch.no_grad():
        gp, gcoors = gpu.voxelize([torch.from_numpy(pts).to(dev)])
        gvf, gvc = gpu.pts_voxel_encoder(gp, gcoors)
        np.testing.assert_array_equal(gvc.cpu().numpy(), vc)
        np.testing.assert_allclose(gvf.cpu().numpy(), vf, rtol=1e-4, atol=1e-5)
        bev = gpu.pts_middle_encoder(gvf, gvc, 1)
    # same voxel features on both sides -> the sparse encoder must agree exactly
    bev_ref = pipeline.sparse_encoder(cpu.pts_middle_encoder, gvf.cpu().numpy(), vc, 1)
    assert np.array_equal(bev.cpu().numpy(), bev_ref)
    expect = (1, 256, 200, 176) if "kitti" in name else (1, 256, 192, 192)
    assert tuple(bev.shape) == expect
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    with torch.no_grad():
        res = gpu.simple_test(None, [torch.from_numpy(pts).to(dev)], metas)
    out = res[0]["pts_bbox"] if "pts_bbox" in res[0] else res[0]
    assert set(out.keys()) == {"boxes_3d", "scores_3d", "labels_3d"}
    assert out["boxes_3d"].tensor.shape[1] == (7 if "kitti" in name or "waymo" in name else 9)


def test_batch_of_two_frames(setup, dev):
    """bs = 2: per-sample voxelization with batch ids, one sparse encoder pass over both samples, per-sample attention."""
    cpu, gpu, pts = setup
    frames = [pts, S.nuscenes_sweep(2001, 20000)]
    vf, coors = pipeline.voxel_features(cpu, frames)
    bev_ref = pipeline.sparse_encoder(cpu.pts_middle_encoder, vf, coors, 2)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes), dict(box_type_3d=LiDARInstance3DBoxes)]
    rec = []
    hooks = [st.register_forward_pre_hook(lambda m, a: rec.append((a[1].detach().clone().cpu().numpy(),
                                                                  a[2].detach().clone().cpu().numpy().reshape(2, 200, -1))))
             for st in gpu.bbox_head.head_series_lidar]
    with torch.no_grad():
        bev = gpu.extract_bev([torch.from_numpy(f).to(dev) for f in frames])
        assert np.array_equal(bev.cpu().numpy(), bev_ref)
        feats = gpu.pts_neck(gpu.pts_backbone(bev))
        logits, boxes = gpu.bbox_head(None, feats, metas)
        res = gpu.forward(return_loss=False, points=[[torch.from_numpy(f).to(dev) for f in frames]], img_metas=[metas])
    for h in hooks:
        h.remove()
    assert boxes.shape == (5, 2, 200, 10) and len(res) == 2
    ref_logits, ref_boxes = pipeline.head_forward(cpu.bbox_head, None, [f.cpu() for f in feats], metas, stage_inputs=rec)
    np.testing.assert_allclose(boxes.cpu().numpy(), ref_boxes.numpy(), rtol=0, atol=1e-4)
    # each sample of the batch equals the same frame run alone (the first stages are tight; see test_oracle_pinned)
    with torch.no_grad():
        # (the lone sample in the layout the batch has: an NCHW-contiguous copy would send the proposal generator down its
        # torch route, whose channel sum adds in another order -- 1e-6 on the proposals, 1e-4 after a random-weight stage)
        l1, b1 = gpu.bbox_head(None, [f[1:2].contiguous(memory_format=torch.channels_last) for f in feats], metas[1:])
    np.testing.assert_allclose(boxes[0, 1].cpu().numpy(), b1[0, 0].cpu().numpy(), rtol=0, atol=1e-4)


def test_checkpoint_round_trip(setup, dev):
    cpu, gpu, pts = setup
    import io
    buf = io.BytesIO()
    torch.save({"state_dict": gpu.state_dict()}, buf)
    buf.seek(0)
    sd = torch.load(buf, weights_only=True)["state_dict"]
    fresh = workloads.build("srfdet_voxel_nusc_L", 200).eval().to(dev)
    missing, unexpected = fresh.load_state_dict(sd, strict=True)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    p = torch.from_numpy(pts).to(dev)
    gpu.bbox_head.test_cfg = dict(gpu.bbox_head.test_cfg, score_thr=0.02)
    fresh.bbox_head.test_cfg = dict(fresh.bbox_head.test_cfg, score_thr=0.02)
    try:
        with torch.no_grad():
            a = gpu.simple_test(None, [p], metas)[0]["pts_bbox"]
            b = fresh.simple_test(None, [p], metas)[0]["pts_bbox"]
    finally:
        gpu.bbox_head.test_cfg = dict(gpu.bbox_head.test_cfg, score_thr=0.1)
    assert a["scores_3d"].numel() > 5
    assert torch.equal(a["labels_3d"], b["labels_3d"])
    torch.testing.assert_close(a["boxes_3d"].tensor, b["boxes_3d"].tensor, rtol=0, atol=1e-4)


def test_lc_frame_eager_graphs_and_overlap_agree(dev):
    """LC configuration end to end (small camera images so the test stays quick): eager, hipGraph tail + image-branch
    graph on the main stream, and the image graph on a side stream must give the same detections; the fused dense route
    (BN + ReLU pass, OSA concat convolution) must agree with the plain module chain."""
    import copy
    torch.manual_seed(2)
    cpu = workloads.build("srfdet_voxel_nusc_LC", 48).eval()
    _randomize_bn(cpu, 2)
    # random weights give sigmoid scores around 0.5 at best: lower the threshold so that the NMS really has work to do
    cpu.bbox_head.test_cfg = dict(cpu.bbox_head.test_cfg, score_thr=0.02)
    eager = copy.deepcopy(cpu).to(dev)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in S.camera_rig(f=1266.0 * 256 / 1600, cx=128.0, cy=80.0)])]
    img = torch.from_numpy(S.camera_images(3000, h=160, w=256)).to(dev)
    frames = [torch.from_numpy(S.nuscenes_sweep(2000 + i, 12000)).to(dev) for i in range(2)]
    def decoded_eager(p, mt):
        mt = copy.deepcopy(mt)
        with torch.no_grad():
            img_feats, pt_feats = eager.extract_feat(img, [p], mt)
            return eager.bbox_head.decode(*eager.bbox_head(img_feats, pt_feats, mt))

    def decoded_graph(g, p, mt):
        """runs the frame through the graphs, returns the detections and the pre-NMS tensors the graph produced (the parity
        contract is on those; which of two nearly tied boxes survives the NMS may flip on the last bits)"""
        with torch.no_grad():
            det = g.simple_test(img, [p], copy.deepcopy(mt))[0]["pts_bbox"]
        e = g._graphed_frame.entry if g._graphed_frame is not None else list(g._graphed_tail.entries.values())[-1]
        return det, e["scores"].clone(), e["boxes"].clone()

    def close(got_s, got_b, want):
        torch.testing.assert_close(got_s, want[0], rtol=0, atol=1e-5)
        torch.testing.assert_close(got_b, want[1], rtol=2e-5, atol=1e-4)

    with torch.no_grad():
        ref = [decoded_eager(p, metas) for p in frames]
        ref_det = [eager.simple_test(img, [p], copy.deepcopy(metas))[0]["pts_bbox"] for p in frames]
        assert all(r["scores_3d"].numel() > 5 for r in ref_det), "the test needs detections"
        # plain module chain for the image branch (no fused passes): autocast-free fp32, grad mode on disables fusion
        feats_fused = eager.extract_img_feat(img, copy.deepcopy(metas))
    with torch.enable_grad():
        feats_plain = eager.extract_img_feat(img, copy.deepcopy(metas))
    for a, b in zip(feats_fused, feats_plain):
        torch.testing.assert_close(a, b.detach(), rtol=1e-4, atol=1e-4)
    metas2 = [dict(box_type_3d=LiDARInstance3DBoxes,
                   lidar2img=[m for m in S.camera_rig(f=1266.0 * 256 / 1600 * 1.3, cx=120.0, cy=70.0, cam_h=1.2)])]
    ref2 = decoded_eager(frames[0], metas2)
    assert (ref2[1] - ref[0][1]).abs().max() > 1e-3, "the second calibration must change the result"
    for overlap, whole in ((False, True), (True, True), (False, False)):
        g = copy.deepcopy(cpu).to(dev).enable_hip_graphs(img_overlap=overlap, whole_frame=whole)
        with torch.no_grad():
            g.simple_test(img, [frames[0]], copy.deepcopy(metas))  # first call: eager + capture; replays from here on
        for rep in range(2):
            for p, want, want_det in zip(frames, ref, ref_det):
                det, gs, gb = decoded_graph(g, p, metas)
                close(gs, gb, want)
                assert abs(det["scores_3d"].numel() - want_det["scores_3d"].numel()) <= 2
        # new calibration on a later frame (real data: lidar2img changes every sample): the replay must use it
        _, gs, gb = decoded_graph(g, frames[0], metas2)
        close(gs, gb, ref2)
        _, gs, gb = decoded_graph(g, frames[0], metas)
        close(gs, gb, ref[0])


def test_whole_frame_graph_equals_eager(setup, dev):
    """GraphedFrame: voxelization + sparse encoder + tail as ONE hipGraph with capacity-padded levels.  Must reproduce the
    eager detections on replay, on a sweep with fewer points (filler rows), after a sweep with MORE points than the
    point buffer (eager fallback + recapture), and after a capacity overflow (forced with a tiny headroom)."""
    import copy
    from srfdet3d_amd.graphs import GraphedFrame
    cpu, gpu, _ = setup
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]

    def same(a, b):
        assert b["scores_3d"].numel() > 5, "the test needs detections to compare"
        assert torch.equal(a["labels_3d"], b["labels_3d"])
        torch.testing.assert_close(a["scores_3d"], b["scores_3d"], rtol=0, atol=1e-5)
        torch.testing.assert_close(a["boxes_3d"].tensor, b["boxes_3d"].tensor, rtol=2e-5, atol=1e-4)

    gpu = copy.deepcopy(gpu)
    gpu.bbox_head.test_cfg = dict(gpu.bbox_head.test_cfg, score_thr=0.02)   # so that the NMS has candidates
    g = copy.deepcopy(gpu).enable_hip_graphs()
    assert g._graphed_frame is not None
    sweeps = [S.nuscenes_sweep(2000, 30000), S.nuscenes_sweep(2001, 30000), S.nuscenes_sweep(2002, 24000),
              S.nuscenes_sweep(2003, 45000), S.nuscenes_sweep(2000, 30000)]
    for sw in sweeps:
        p = torch.from_numpy(sw).to(dev)
        with torch.no_grad():
            same(g.simple_test(None, [p], metas)[0]["pts_bbox"], gpu.simple_test(None, [p], metas)[0]["pts_bbox"])
    st = g._graphed_frame.stats
    assert st["replays"] >= 3 and st["captures"] == 2 and st["eager"] == 2, st   # first frame + the 45k-point sweep
    # the BEV map of the static path is bit-identical to the dynamic one (padding rows are never read)
    with torch.no_grad():
        p = torch.from_numpy(sweeps[0]).to(dev)
        bev = gpu.extract_bev([p])
        far = torch.full((4096, p.shape[1]), 1.0e6, device=dev)
        caps = dict(g._graphed_frame.entry["caps"])
        bev_s, counts = gpu.extract_bev_static(torch.cat([p, far]), caps)
        assert torch.equal(bev, bev_s)
        assert all(int(c[1].item()) <= c[2] for c in counts)
        # ... and does not depend on whether the rulebooks were built by the index-only pass on a second stream (the default
        # of the static path) or in line with the convolutions
        import os
        os.environ["SRF_SPARSE_INDEX_STREAM"] = "0"
        try:
            bev_1, _ = gpu.extract_bev_static(torch.cat([p, far]), caps)
        finally:
            del os.environ["SRF_SPARSE_INDEX_STREAM"]
        assert torch.equal(bev_1, bev_s)
    # overflow: capacities sized for a sparse sweep, then a dense one
    old = GraphedFrame.HEADROOM
    GraphedFrame.HEADROOM = 1.0
    try:
        g2 = copy.deepcopy(gpu).enable_hip_graphs()
        small = torch.from_numpy(S.nuscenes_sweep(2005, 12000)).to(dev)
        big = torch.cat([torch.from_numpy(S.nuscenes_sweep(2006, 6000)), torch.from_numpy(S.nuscenes_sweep(2007, 6000))]).to(dev)
        with torch.no_grad():
            same(g2.simple_test(None, [small], metas)[0]["pts_bbox"], gpu.simple_test(None, [small], metas)[0]["pts_bbox"])
            for _ in range(2):
                same(g2.simple_test(None, [big], metas)[0]["pts_bbox"], gpu.simple_test(None, [big], metas)[0]["pts_bbox"])
    finally:
        GraphedFrame.HEADROOM = old


@pytest.mark.parametrize("name,sweep,npts,np_", [("srfdet_voxel_kitti_L", "kitti_sweep", 17000, 100),
                                                 ("srfdet_dvoxel_waymo_L", "waymo_sweep", 60000, 64)])
def test_whole_frame_graph_dynamic_voxel_configs(name, sweep, npts, np_, dev):
    """GraphedFrame on the dynamic-voxelization configs (DynamicVFECustom at a fixed voxel capacity): the replayed frame
    must reproduce the eager pre-NMS tensors on several sweeps, including one with fewer points than the buffer."""
    import copy
    torch.manual_seed(3)
    cpu = workloads.build(name, np_).eval()
    _randomize_bn(cpu, 3)
    cpu.bbox_head.test_cfg = dict(cpu.bbox_head.test_cfg, score_thr=0.02)
    eager = copy.deepcopy(cpu).to(dev)
    g = copy.deepcopy(cpu).to(dev).enable_hip_graphs()
    assert g._graphed_frame is not None
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    seed = 1000 if "kitti" in name else 5000
    gen = getattr(S, sweep)
    sweeps = [gen(seed, npts), gen(seed + 1, npts), gen(seed + 2, int(npts * 0.8)), gen(seed, npts)]
    for i, sw in enumerate(sweeps):
        p = torch.from_numpy(sw).to(dev)
        with torch.no_grad():
            pf = eager.extract_point_features([p])
            want_s, want_b = eager.bbox_head.decode(*eager.bbox_head(None, pf, metas))
            det = g.simple_test(None, [p], metas)
        if i == 0:
            continue  # first call: eager pass + capture
        e = g._graphed_frame.entry
        torch.testing.assert_close(e["scores"], want_s, rtol=0, atol=1e-5)
        torch.testing.assert_close(e["boxes"], want_b, rtol=2e-5, atol=1e-4)
    st = g._graphed_frame.stats
    assert st["replays"] == 3 and st["captures"] == 1, st


@pytest.mark.parametrize("name,n_cam,sweep,npts", [("srfdet_dvoxel_nusc_L", 0, "nuscenes_sweep", 20000),
                                                   ("srfdet_pillar_v299_nusc_LC", 6, "nuscenes_sweep", 20000),
                                                   ("srfdet_voxel_r50_nusc_LC", 6, "nuscenes_sweep", 20000),
                                                   ("srfdet_pillar_r50_nusc_LC", 6, "nuscenes_sweep", 20000),
                                                   ("srfdet_voxel_kitti_LC", 1, "kitti_sweep", 17000),
                                                   ("srfdet_dvoxel_waymo_LC", 5, "waymo_sweep", 40000)])
def test_remaining_reference_configs_run_and_graphs_agree(name, n_cam, sweep, npts, dev):
    """The other configs of the reference (dynamic-voxel nuScenes, pillar + VoVNet, ResNet-50 image backbones, KITTI with its
    single camera, Waymo LC with its DCNv2 ResNet-101): one frame end to end, eager and through the hipGraphs, same pre-NMS
    tensors."""
    import copy
    torch.manual_seed(4)
    cpu = workloads.build(name, 32).eval()
    _randomize_bn(cpu, 4)
    eager = copy.deepcopy(cpu).to(dev)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    img = None
    if n_cam:
        rig = S.camera_rig(n_cam=n_cam, f=1266.0 * 256 / 1600, cx=128.0, cy=80.0)
        metas[0]["lidar2img"] = rig[0] if n_cam == 1 else [m for m in rig]
        img = torch.from_numpy(S.camera_images(3000, n_cam=n_cam, h=160, w=256)).to(dev)
        if n_cam == 1:
            img = img[:, 0]          # KITTI: (B, 3, H, W)
    pts = [torch.from_numpy(getattr(S, sweep)(7000 + i, npts)).to(dev) for i in range(2)]
    with torch.no_grad():
        want = []
        for p in pts:
            mt = copy.deepcopy(metas)
            f_img, f_pt = eager.extract_feat(img, [p], mt)
            s, b = eager.bbox_head.decode(*eager.bbox_head(f_img, f_pt, mt))
            assert torch.isfinite(s).all() and torch.isfinite(b).all()
            want.append((s, b))
        res = eager.simple_test(img, [pts[0]], copy.deepcopy(metas))
        assert len(res) == 1
    g = copy.deepcopy(cpu).to(dev).enable_hip_graphs()
    with torch.no_grad():
        g.simple_test(img, [pts[0]], copy.deepcopy(metas))                     # eager pass + capture
        for i in (1, 0, 1):
            g.simple_test(img, [pts[i]], copy.deepcopy(metas))
            e = g._graphed_frame.entry if g._graphed_frame is not None else list(g._graphed_tail.entries.values())[-1]
            # five free-running random-weight stages amplify the last-bit differences between MIOpen's eager and captured
            # algorithm choices; a broken graph is off by 1e-2 and more
            torch.testing.assert_close(e["scores"], want[i][0], rtol=0, atol=2e-4)
            # boxes: all but a stray element within 1e-3 (one size entry -- exp() of a free-running delta -- of the Waymo LC
            # config moves by 1e-3..3e-3 from run to run), every element within 1e-2
            tight = torch.isclose(e["boxes"], want[i][1], rtol=1e-3, atol=2e-3)
            assert tight.float().mean().item() >= 0.99, f"{(~tight).sum().item()} of {tight.numel()} box entries differ"
            torch.testing.assert_close(e["boxes"], want[i][1], rtol=1e-2, atol=2e-3)
