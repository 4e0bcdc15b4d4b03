"""Training-side checks: gradient of the RoI gather, 3-D IoU, OTA assignment invariants, one
LiDAR+camera training step with the LiDAR branch frozen as tools/train.py:221-276 does, and a 2-rank DDP step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from srfdet3d_amd import ops, synthetic as S, workloads
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes
from srfdet3d_amd.plugin import training

pytestmark = pytest.mark.gpu


def _torch_roi_align(feat, rois, scale, P=7, sr=2):
    """differentiable restatement of RoIAlign(avg, aligned) in torch ops (double precision) for gradient checking."""
    N, C, H, W = feat.shape
    out = []
    for r in rois:
        n = int(r[0])
        x1, y1, x2, y2 = [r[i] * scale - 0.5 for i in range(1, 5)]
        bw, bh = (x2 - x1) / P, (y2 - y1) / P
        idx = torch.arange(P * sr, dtype=feat.dtype, device=feat.device)
        ys = y1 + (idx // sr) * bh + ((idx % sr) + 0.5) * bh / sr
        xs = x1 + (idx // sr) * bw + ((idx % sr) + 0.5) * bw / sr
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        valid = ~((yy < -1) | (yy > H) | (xx < -1) | (xx > W))
        yy, xx = yy.clamp(min=0), xx.clamp(min=0)
        y0, x0 = yy.floor().long().clamp(max=H - 1), xx.floor().long().clamp(max=W - 1)
        y1i, x1i = (y0 + 1).clamp(max=H - 1), (x0 + 1).clamp(max=W - 1)
        yy = torch.where(y0 >= H - 1, y0.to(feat.dtype), yy)
        xx = torch.where(x0 >= W - 1, x0.to(feat.dtype), xx)
        ly, lx = yy - y0, xx - x0
        f = feat[n]
        v = ((1 - ly) * (1 - lx) * f[:, y0, x0] + (1 - ly) * lx * f[:, y0, x1i] + ly * (1 - lx) * f[:, y1i, x0] + ly * lx * f[:, y1i, x1i])
        v = v * valid
        out.append(v.view(C, P, sr, P, sr).mean(dim=(2, 4)))
    return torch.stack(out)


def test_roi_extract_backward_matches_autograd(dev):
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(2, 6, s, s + 2, generator=g) for s in (24, 12, 6, 3)]
    rois = torch.tensor([[0, 10.3, 20.1, 80.7, 90.2], [1, -5, -8, 30, 25], [0, 40, 30, 170, 150], [1, 2, 3, 190, 185],
                         [0, 150, 150, 151, 152]], dtype=torch.float32)
    strides = [8, 16, 32, 64]
    lv = ops.roi_extract([f.to(dev) for f in feats], rois.to(dev), strides, return_levels=True)[1].cpu().tolist()
    assert len(set(lv)) >= 2
    w = torch.randn(5, 6, 7, 7, generator=g)
    fd = [f.double().requires_grad_(True) for f in feats]
    ref = torch.stack([_torch_roi_align(fd[l], rois[i:i + 1].double(), 1.0 / strides[l])[0] for i, l in enumerate(lv)])
    (ref * w.double()).sum().backward()
    for layout in ("nchw", "cl"):
        fg = [f.to(dev).requires_grad_(True) for f in feats]
        fin = [f.contiguous(memory_format=torch.channels_last) for f in fg] if layout == "cl" else fg
        out = ops.roi_extract_autograd(fin, rois.to(dev), strides)
        torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-5, atol=1e-5)
        (out * w.to(dev)).sum().backward()
        for a, b in zip(fg, fd):
            want = b.grad if b.grad is not None else torch.zeros_like(b)  # a level no RoI maps to gets a zero gradient
            torch.testing.assert_close(a.grad.cpu().double(), want, rtol=1e-4, atol=1e-5)


def test_iou3d_against_axis_aligned_cases(dev):
    a = torch.tensor([[0, 0, 0, 2, 4, 1, 0.0], [0, 0, 0, 2, 4, 1, np.pi / 2], [10, 10, 0, 1, 1, 1, 0.3]], device=dev)
    b = torch.tensor([[0, 0, 0, 2, 4, 1, 0.0], [1, 0, 0.5, 2, 4, 1, 0.0], [0, 0, 0, 4, 2, 1, 0.0]], device=dev)
    iou = training.bbox_overlaps_3d(a, b).cpu()
    assert abs(iou[0, 0] - 1.0) < 1e-5
    # shifted by 1 in x and 0.5 in z: bev overlap 1x4, z overlap 0.5 -> 2 / (8 + 8 - 2)
    assert abs(iou[0, 1] - 2.0 / 14.0) < 1e-5
    # 2x4 rotated by 90 deg == 4x2 unrotated
    assert abs(iou[1, 2] - 1.0) < 1e-4
    assert iou[2].abs().max() == 0


def _gt(dev, n=12, seed=0):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-40, 40, (n, 2))
    z = rng.uniform(-2.5, -0.5, (n, 1))
    size = rng.uniform([1.5, 3.5, 1.4], [2.2, 5.0, 2.0], (n, 3))
    yaw = rng.uniform(-np.pi, np.pi, (n, 1))
    vel = rng.normal(0, 1, (n, 2))
    t = torch.tensor(np.concatenate([xy, z, size, yaw, vel], 1), dtype=torch.float32, device=dev)
    return LiDARInstance3DBoxes(t, box_dim=9), torch.from_numpy(rng.integers(0, 10, n)).to(dev)


def test_ota_assignment_invariants(dev):
    m = workloads.model_cfg("srfdet_voxel_nusc_L")
    asg = training.OTAssignerSRFDet(**{k: v for k, v in m.train_cfg.assigner.items() if k != "type"})
    gtb, gtl = _gt(dev)
    gts = torch.cat((gtb.gravity_center, gtb.tensor[:, 3:]), dim=1)
    g = torch.Generator().manual_seed(1)
    P = 300
    pred = torch.zeros(1, P, 10)
    pred[0, :, :3] = torch.rand(P, 3, generator=g) * torch.tensor([110.4, 110.4, 8.0]) + torch.tensor([-55.2, -55.2, -5.0])
    pred[0, :, 3:6] = torch.log(torch.tensor([1.9, 4.2, 1.7]))
    pred[0, :, 7] = 1.0
    logits = torch.randn(1, P, 10, generator=g)
    # twelve predictions sit on the ground truth (centre + 5 cm, same size and heading, confident right class)
    pred[0, :12, :3] = gts[:, :3].cpu() + 0.05
    pred[0, :12, 3:6] = gts[:, 3:6].cpu().log()
    pred[0, :12, 6], pred[0, :12, 7] = gts[:, 6].cpu().sin(), gts[:, 6].cpu().cos()
    logits[0, :12] = -4.0
    logits[0, torch.arange(12), gtl.cpu()] = 4.0
    out = dict(pred_boxes=pred.to(dev), pred_logits=logits.to(dev))
    for head_idx in (1, 5):
        fg, j = asg(out, [gts], [gtl], head_idx)[0]
        assert fg.dtype == torch.bool and fg.sum() == j.numel() >= 12
        assert set(j.cpu().tolist()) == set(range(12)), "every ground-truth box gets at least one prediction"
        assert fg[:12].all(), "the predictions placed on the ground truth are selected"
    fg, j = asg(out, [gts[:0]], [gtl[:0]], 5)[0]
    assert fg.sum() == 0 and j.numel() == 0


def _lc_inputs(dev, seed=0):
    pts = torch.from_numpy(S.nuscenes_sweep(2000 + seed, 8000)).to(dev)
    img = torch.from_numpy(S.camera_images(3000 + seed, h=128, w=224)).to(dev)
    gtb, gtl = _gt(dev, seed=seed)
    metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in S.camera_rig(f=177.0, cx=112.0, cy=64.0)])]
    return dict(img=img, points=[pts], img_metas=metas, gt_bboxes_3d=[gtb], gt_labels_3d=[gtl])


def _lc_model(dev, P=64):
    torch.manual_seed(0)
    m = workloads.build("srfdet_voxel_nusc_LC", P, train=True)
    training.freeze_lidar_components(m)
    return m.to(dev).train()


def test_lc_training_step(dev):
    model = _lc_model(dev)
    assert not model.pts_middle_encoder.training and model.bbox_head.training
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01)
    losses = model(return_loss=True, **_lc_inputs(dev))
    assert set(losses) == {"loss_cls", "loss_bbox"} | {f"s.{i}.{n}" for i in range(4) for n in ("loss_cls", "loss_bbox")}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    named = dict(model.named_parameters())
    for k in ("bbox_head.head_series_lidar.0.output_fused_proj.weight", "bbox_head.head_series_lidar.4.bboxes_delta_lidar.weight",
              "bbox_head.img_convs.0.weight", "img_neck.fpn_convs.0.conv.weight", "img_backbone.stage5.OSA5_3.concat.OSA5_3_concat/conv.weight",
              "bbox_head.dpg_fc1_img.weight", "bbox_head.init_proposal_boxes.weight"):
        g = named[k].grad
        assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0, k
    assert all(p.grad is None for n, p in named.items() if n.startswith("pts_"))
    before = named["bbox_head.head_series_lidar.0.output_fused_proj.weight"].detach().clone()
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 35.0)
    opt.step()
    assert not torch.equal(before, named["bbox_head.head_series_lidar.0.output_fused_proj.weight"])


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks share the one GPU of the box -> gloo
    dev = torch.device("cuda:0")
    model = _lc_model(dev, P=32)
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
    losses = ddp(return_loss=True, **_lc_inputs(dev, seed=rank))  # each rank trains on its own frame
    sum(losses.values()).backward()
    g = dict(model.named_parameters())["bbox_head.head_series_lidar.0.output_fused_proj.weight"].grad
    # by value (numpy), not a torch tensor: torch shares CPU tensors through /dev/shm files that vanish when this
    # process exits before the parent has unpickled them
    q.put((rank, float(sum(losses.values())), g.detach().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_average_gradients(dev):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.isfinite(res[0][1]) and np.isfinite(res[1][1]) and res[0][1] != res[1][1]  # different frames
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1.3e-6, atol=1e-5)  # gradients all-reduced: identical on both ranks


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("N,Cin,Cout,H,W,bias", [(2, 64, 96, 13, 21, True), (3, 192, 192, 29, 50, False), (1, 256, 128, 58, 100, True)])
def test_training_conv3x3_on_the_winograd_kernel_matches_autograd(dev, N, Cin, Cout, H, W, bias, channels_last):
    """train_conv._Wino43Conv (forward and data gradient on srf_wino43, weight gradient through aten.convolution_backward on the
    channels-last operands) against torch's own autograd through F.conv2d in float64: 1e-4 relative to each tensor's maximum, as
    tests/test_gpu_spconv_bwd.py holds the sparse convolution's gradients."""
    import torch.nn.functional as F
    from srfdet3d_amd import train_conv
    g = torch.Generator().manual_seed(Cin + H)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=bias).to(dev)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    gy = torch.randn(N, Cout, H, W, generator=g).to(dev)
    assert train_conv.eligible(conv, x)
    y = train_conv.conv2d(conv, x)
    assert y.grad_fn is not None                                  # on the autograd tape
    # the caller's memory format is kept (the NCHW-in / NCHW-out branch had no test): channels-last in -> channels-last out,
    # NCHW-contiguous in -> NCHW-contiguous out, and the same for the input gradient
    assert (y.stride(1) == 1) if channels_last else y.is_contiguous()
    y.backward(gy)
    assert (x.grad.stride(1) == 1) if channels_last else x.grad.is_contiguous()
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = conv.weight.detach().double().cpu().requires_grad_(True)
    bd = conv.bias.detach().double().cpu().requires_grad_(True) if bias else None
    yd = F.conv2d(xd, wd, bd, padding=1)
    yd.backward(gy.double().cpu())
    for got, want in ((y, yd), (x.grad, xd.grad), (conv.weight.grad, wd.grad)) + (((conv.bias.grad, bd.grad),) if bias else ()):
        err = (got.detach().double().cpu() - want.detach()).abs().max().item()
        assert err <= 1e-4 * want.detach().abs().max().item(), (err, want.detach().abs().max().item())
    # not taken without autograd or when the switch is off
    with torch.no_grad():
        assert not train_conv.eligible(conv, x)


def test_training_depthwise_and_1x1_routes_match_autograd(dev):
    """train_conv.conv2d: the depthwise stride-2 stair convolutions on torch's native kernels in both directions, the 1x1 layers as
    a GEMM on the (pixels, channels) view, eval-mode BatchNorm as an affine map -- gradients against float64 autograd."""
    import torch.nn.functional as F
    from srfdet3d_amd import train_conv
    g = torch.Generator().manual_seed(9)
    for conv in (torch.nn.Conv2d(64, 64, 3, stride=2, padding=1, groups=64, bias=False), torch.nn.Conv2d(96, 160, 1, bias=True)):
        conv = conv.to(dev)
        x = torch.randn(2, conv.in_channels, 21, 34, generator=g).to(dev)
        if conv.kernel_size == (1, 1):
            # the GEMM route is taken for channels-last tensors only (an NCHW input silently took torch's convolution)
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        if conv.kernel_size == (1, 1):
            assert train_conv.eligible_1x1(conv, x)
        else:
            assert train_conv.eligible_depthwise(conv, x)
        bn = torch.nn.BatchNorm2d(conv.out_channels).to(dev).eval()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
        y = train_conv.bn_eval(bn, train_conv.conv2d(conv, x))
        gy = torch.randn(y.shape, generator=g).to(dev)
        y.backward(gy)
        xd = x.detach().double().cpu().requires_grad_(True)
        c64, b64 = torch.nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, groups=conv.groups,
                                   bias=conv.bias is not None).double(), torch.nn.BatchNorm2d(conv.out_channels).double().eval()
        c64.load_state_dict({k: v.double().cpu() for k, v in conv.state_dict().items()})
        b64.load_state_dict({k: (v.double().cpu() if v.is_floating_point() else v.cpu()) for k, v in bn.state_dict().items()})
        yd = b64(c64(xd))
        yd.backward(gy.double().cpu())
        pairs = [(y, yd), (x.grad, xd.grad), (conv.weight.grad, c64.weight.grad), (bn.weight.grad, b64.weight.grad), (bn.bias.grad, b64.bias.grad)]
        for got, want in pairs:
            err = (got.detach().double().cpu() - want.detach()).abs().max().item()
            assert err <= 1e-4 * max(want.detach().abs().max().item(), 1e-3), (err,)


def test_train_mode_batchnorm_never_sees_a_channels_last_tensor(dev):
    """ADVICE r3 / the round-3 crash: MIOpen's TRAINING batch-norm segfaults (host side) on a channels-last tensor of batch size 1
    (tools/bn_channels_last_probe.py: (1, 128, 23, 23) and (1, 64, 23, 23) crash; 24 x 24 / 46 x 46, batch size 2, NCHW, eval mode do not).  The guard sits where every
    module-path BatchNorm passes (`train_conv.bn_train_input`, used by ConvModule, run_sequential and conv_bn_act): the exact
    failing chain -- channels-last input, 3x3 ConvModule with BN in train mode, two stride-2 ConvModules with BN in train mode, batch
    size 1 -- runs forward and backward and matches the same modules on an NCHW-contiguous input."""
    from srfdet3d_amd import train_conv
    from srfdet3d_amd.compat.cnn import ConvModule
    torch.manual_seed(0)
    norm = dict(type="BN2d", eps=1e-3, momentum=0.01)
    mods = torch.nn.Sequential(ConvModule(128, 128, 3, padding=1, norm_cfg=norm), ConvModule(128, 128, 3, stride=2, padding=1, norm_cfg=norm),
                               ConvModule(128, 128, 3, stride=2, padding=1, norm_cfg=norm, act_cfg=None)).to(dev).train()
    x = torch.randn(1, 128, 92, 92, device=dev)
    xc = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xn = x.clone().requires_grad_(True)
    seen = []
    orig = torch.nn.functional.batch_norm

    def spy(t, *a, **k):
        seen.append((tuple(t.shape), t.is_contiguous(), bool(a[4] if len(a) > 4 else k.get("training", False))))
        return orig(t, *a, **k)

    torch.nn.functional.batch_norm = spy
    try:
        yc = mods(xc)
        yc.square().mean().backward()
    finally:
        torch.nn.functional.batch_norm = orig
    assert len(seen) == 3 and all(contig and training for _, contig, training in seen), seen
    assert seen[-1][0] == (1, 128, 23, 23)
    state = {k: v.clone() for k, v in mods.state_dict().items()}
    gc = xc.grad.clone()
    for m in mods.modules():          # same running statistics for the second pass
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    yn = mods(xn)
    yn.square().mean().backward()
    torch.testing.assert_close(yc, yn, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gc, xn.grad, rtol=1e-3, atol=1e-5)
    # the helper itself: only a train-mode BatchNorm triggers the copy
    bn = torch.nn.BatchNorm2d(128).to(dev)
    t = torch.randn(1, 128, 23, 23, device=dev).contiguous(memory_format=torch.channels_last)
    assert train_conv.bn_train_input(bn.train(), t).is_contiguous()
    assert train_conv.bn_train_input(bn.eval(), t) is t


@pytest.mark.parametrize("k,N,Cin,Cout,H,W,relu,bias", [(3, 2, 192, 192, 29, 50, True, False), (3, 1, 160, 224, 13, 21, False, True),
                                                        (1, 2, 1472, 768, 29, 25, True, False), (1, 3, 64, 96, 17, 9, True, True)])
def test_fused_conv_evalbn_relu_training_node_matches_autograd(dev, k, N, Cin, Cout, H, W, relu, bias):
    """train_conv._ConvAffineRelu (conv + eval-mode BatchNorm + ReLU: one forward launch, `srf_nhwc_affine_relu_bwd` + dgrad + wgrad
    backward) against torch's autograd through conv2d -> batch_norm(eval) -> relu in float64"""
    from torch import nn
    from srfdet3d_amd import dense, train_conv
    g = torch.Generator().manual_seed(k * 100 + Cin)
    torch.manual_seed(k * 100 + Cin)   # (the convolution's default init draws from the global generator: without this the data depend on
    # which tests ran before, and with them how many outputs sit within rounding of the ReLU's zero)
    conv = nn.Conv2d(Cin, Cout, k, padding=k // 2, bias=bias).to(dev)
    bn = nn.BatchNorm2d(Cout, eps=1e-3).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn.eval()
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(N, Cout, H, W, generator=g).to(dev)
    assert train_conv.fused_eligible(conv, bn, x)
    y = dense.conv_bn_act(conv, bn, relu, x)
    assert type(y.grad_fn).__name__ == "_ConvAffineReluBackward" and y.stride(1) == 1
    y.backward(gy)
    got = [y.detach(), x.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad] + ([conv.bias.grad] if bias else [])
    # float64 reference through torch's own operators
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True) if bias else None
    gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    z = F.conv2d(xd, wd, bd, padding=k // 2)
    u = F.batch_norm(z, bn.running_mean.double(), bn.running_var.double(), gam, bet, False, 0.0, bn.eps)
    yr = torch.relu(u) if relu else u
    yr.backward(gy.double())
    ref = [yr.detach(), xd.grad, wd.grad, gam.grad, bet.grad] + ([bd.grad] if bias else [])
    names = ["y", "dx", "dW", "dgamma", "dbeta", "dbias"]
    for name, a, b in zip(names, got, ref):
        scale = float(b.abs().max())
        err = float((a.double() - b).abs().max()) / max(scale, 1e-30)
        # the ReLU mask of an element within rounding of zero may differ between the f32 kernel and float64: a handful of elements
        tol = 2e-3 if name in ("dx",) else 5e-4
        assert err < tol, (name, err)
    # the switch: SRF_TRAIN_FUSED=0 runs the same layer as separate autograd nodes
    os.environ["SRF_TRAIN_FUSED"] = "0"
    try:
        y2 = dense.conv_bn_act(conv, bn, relu, x)
        assert type(y2.grad_fn).__name__ != "_ConvAffineReluBackward"
        assert float((y2 - got[0]).abs().max()) / float(got[0].abs().max()) < 1e-4
    finally:
        del os.environ["SRF_TRAIN_FUSED"]


@pytest.mark.parametrize("k,gammas", [(3, (0.0,)), (1, (1e-6, -1e-7)), (3, (0.0, 1e-6))])
def test_zero_and_tiny_gamma_channels_get_the_true_gradient(dev, k, gammas):
    """`_ConvAffineRelu` divides two column sums by s = gamma / sqrt(var + eps); a layer with a zero-initialised / pruned
    (gamma == 0) or nearly dead channel must not lose or blow up that channel's d gamma.  `conv_bn_act` sends such a layer through
    plain autograd (`train_conv.gamma_well_conditioned`); every gradient is compared with float64 autograd, the small channels too."""
    from torch import nn
    from srfdet3d_amd import dense, train_conv
    g = torch.Generator().manual_seed(77 + k)
    Cin, Cout, N, H, W = 64, 96, 2, 17, 23
    conv = nn.Conv2d(Cin, Cout, k, padding=k // 2, bias=False).to(dev)
    bn = nn.BatchNorm2d(Cout, eps=1e-3).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.3 + 0.2)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn.eval()
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(N, Cout, H, W, generator=g).to(dev)
    assert train_conv.fused_eligible(conv, bn, x) and train_conv.gamma_well_conditioned(bn)
    assert type(dense.conv_bn_act(conv, bn, True, x).grad_fn).__name__ == "_ConvAffineReluBackward"
    small = [5 + 7 * i for i in range(len(gammas))]
    with torch.no_grad():                                   # an in-place update, as an optimiser step: the cached verdict goes stale
        for c, v in zip(small, gammas):
            bn.weight[c] = v
            bn.bias[c] = 0.4                                # u = beta > 0 on these channels: the ReLU lets their gradient through
    assert not train_conv.gamma_well_conditioned(bn)
    y = dense.conv_bn_act(conv, bn, True, x)
    assert type(y.grad_fn).__name__ != "_ConvAffineReluBackward"
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    u = F.batch_norm(F.conv2d(xd, wd, None, padding=k // 2), bn.running_mean.double(), bn.running_var.double(), gam, bet, False, 0.0, bn.eps)
    torch.relu(u).backward(gy.double())
    for name, a, b in (("dx", x.grad, xd.grad), ("dW", conv.weight.grad, wd.grad), ("dgamma", bn.weight.grad, gam.grad), ("dbeta", bn.bias.grad, bet.grad)):
        err = float((a.double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err < (2e-3 if name == "dx" else 5e-4), (name, err)
    # the channels in question carry a real gradient (beta > 0 for most pixels' mask), and it is the true one
    for c in small:
        want = float(gam.grad[c])
        assert abs(want) > 1e-3 and abs(float(bn.weight.grad[c]) - want) <= 1e-3 * abs(want) + 1e-4
    # a frozen gamma needs no d gamma: the fused node stays
    bn.weight.requires_grad_(False)
    assert type(dense.conv_bn_act(conv, bn, True, x).grad_fn).__name__ == "_ConvAffineReluBackward"


def test_affine_relu_bwd_kernel(dev):
    g = torch.Generator().manual_seed(23)
    for (N, H, W, C, relu) in ((2, 29, 50, 224, True), (1, 7, 5, 1024, True), (3, 11, 13, 40, False), (1, 1, 3, 4, True)):
        gy = torch.randn(N, H, W, C, generator=g).to(dev)
        y = torch.randn(N, H, W, C, generator=g).to(dev)
        if relu:
            y = torch.relu(y)
        s = (torch.rand(C, generator=g) + 0.5).to(dev)
        gz, sums = ops.nhwc_affine_relu_bwd(gy, y, s, relu)
        gu = gy * (y > 0) if relu else gy
        assert torch.equal(gz, gu * s)
        np.testing.assert_allclose(sums[0].cpu().numpy(), gu.double().sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(sums[1].cpu().numpy(), (gu.double() * y.double()).sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-4)
        gz2, sums2 = ops.nhwc_affine_relu_bwd(gy, y, s, relu)
        assert torch.equal(sums, sums2)      # fixed summation order


@pytest.mark.parametrize("cin,width,cout,L,N,H,W,identity,x_grad", [(64, 32, 96, 3, 2, 21, 34, False, True), (96, 64, 96, 2, 1, 17, 40, True, True),
                                                                      (64, 64, 128, 5, 2, 13, 36, False, False)])
def test_osa_block_as_one_autograd_node_matches_float64(dev, cin, width, cout, L, N, H, W, identity, x_grad):
    """train_conv._OSAChain (VoVNet OSA block, vovnet.py:208-230: L chained conv3x3-BN(eval)-ReLU layers + the concat 1x1 conv-BN-ReLU over
    one channels-last buffer, the two gradients of every layer output added inside `srf_nhwc_affine_relu_bwd2`) against float64 autograd
    through torch's own operators: block output, input gradient, every weight / gamma / beta gradient; and against the per-layer nodes
    (SRF_TRAIN_OSA=0)."""
    from srfdet3d_amd import train_conv
    from srfdet3d_amd.plugin.vovnet import OSAModule
    g = torch.Generator().manual_seed(cin + 7 * L)
    torch.manual_seed(cin + 7 * L)
    blk = OSAModule(cin, width, cout, L, "OSAt_1", identity=identity).to(dev)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.3 + 0.1)
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            elif isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.in_channels * m.kernel_size[0] ** 2)) ** 0.5)
    blk.train()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    x = torch.randn(N, cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(x_grad)
    gy = torch.randn(N, cout, H, W, generator=g).to(dev)
    assert train_conv.osa_eligible(blk, x)
    body = train_conv.osa_chain(blk, x)
    assert type(body.grad_fn).__name__ == "_OSAChainBackward" and body.stride(1) == 1
    y = blk(x)
    y.backward(gy)
    params = [p for p in blk.parameters()]
    got = [y.detach()] + ([x.grad.clone()] if x_grad else []) + [p.grad.clone() for p in params]
    # per-layer nodes
    for p in params:
        p.grad = None
    if x_grad:
        x.grad = None
    os.environ["SRF_TRAIN_OSA"] = "0"
    try:
        assert not train_conv.osa_eligible(blk, x)
        y2 = blk(x)
        y2.backward(gy)
    finally:
        del os.environ["SRF_TRAIN_OSA"]
    per_layer = [y2.detach()] + ([x.grad.clone()] if x_grad else []) + [p.grad.clone() for p in params]
    # float64 through torch's operators
    xd = x.detach().double().requires_grad_(x_grad)
    pd = [p.detach().double().requires_grad_(True) for p in params]
    named = dict(zip([n for n, _ in blk.named_parameters()], pd))
    feats, cur = [xd], xd
    seqs = list(blk.layers) + [blk.concat]
    names = [n for n, _ in blk.named_parameters()]

    def cbr(seq, prefix, inp):
        conv, bn = seq[0], seq[1]
        cname = [n for n in names if n.startswith(prefix) and n.endswith("conv.weight")][0]
        gname = [n for n in names if n.startswith(prefix) and n.endswith("norm.weight")][0]
        bname = [n for n in names if n.startswith(prefix) and n.endswith("norm.bias")][0]
        z = F.conv2d(inp, named[cname], None, padding=conv.kernel_size[0] // 2)
        return torch.relu(F.batch_norm(z, bn.running_mean.double(), bn.running_var.double(), named[gname], named[bname], False, 0.0, bn.eps))

    for i, seq in enumerate(blk.layers):
        cur = cbr(seq, f"layers.{i}.", cur)
        feats.append(cur)
    out = cbr(blk.concat, "concat.", torch.cat(feats, 1))
    fcw, fcb = named["ese.fc.weight"], named["ese.fc.bias"]
    gate = F.relu6(F.conv2d(out.mean(dim=(2, 3), keepdim=True), fcw, fcb) + 3.0) / 6.0
    yr = out * gate
    if identity:
        yr = yr + xd
    yr.backward(gy.double())
    ref = [yr.detach()] + ([xd.grad] if x_grad else []) + [p.grad for p in pd]
    labels = ["y"] + (["dx"] if x_grad else []) + names
    for name, a, b, c in zip(labels, got, ref, per_layer):
        scale = max(float(b.abs().max()), 1e-30)
        err = float((a.double() - b).abs().max()) / scale
        err_pl = float((c.double() - b).abs().max()) / scale
        # (an element within rounding of zero takes the other side of a ReLU than float64 does, and five chained layers pass that on: the
        # per-layer nodes, the accepted route of round 4, show the same error on the same data -- the node must not be worse than they are)
        assert err < 2e-2 and err < max(1.5 * err_pl, 5e-4), (name, err, err_pl)


@pytest.mark.parametrize("N,C,H,W,identity", [(2, 96, 13, 21, True), (3, 64, 9, 40, False), (12, 256, 8, 8, True)])
def test_ese_module_as_one_autograd_node_matches_torch_ops(dev, N, C, H, W, identity):
    """train_conv._ESEApply (pixel mean, gate GEMV, gate multiply + identity; backward: one pass for sum_p g * out, one for
    g * gate + d mean / HW) against the same module through torch's operators in float64 (SRF_TRAIN_ESE=0 runs them in f32)"""
    from srfdet3d_amd import train_conv
    from srfdet3d_amd.plugin.vovnet import eSEModule
    torch.manual_seed(C + H)
    mod = eSEModule(C).to(dev)
    x = (torch.randn(N, C, H, W, device=dev) * 2).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    idt = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) if identity else None
    g = torch.randn(N, C, H, W, device=dev)
    assert train_conv.ese_eligible(mod, x, idt)
    y = mod(x, idt)
    assert type(y.grad_fn).__name__ == "_ESEApplyBackward"
    y.backward(g)
    got = [y.detach(), x.grad, mod.fc.weight.grad, mod.fc.bias.grad] + ([idt.grad] if identity else [])
    xd = x.detach().double().requires_grad_(True)
    wd, bd = mod.fc.weight.detach().double().requires_grad_(True), mod.fc.bias.detach().double().requires_grad_(True)
    idd = idt.detach().double().requires_grad_(True) if identity else None
    gate = F.relu6(F.conv2d(xd.mean(dim=(2, 3), keepdim=True), wd, bd) + 3.0) / 6.0
    yr = xd * gate + (idd if identity else 0.0)
    yr.backward(g.double())
    ref = [yr.detach(), xd.grad, wd.grad, bd.grad] + ([idd.grad] if identity else [])
    for name, a, b in zip(["y", "dx", "dW", "db", "d identity"], got, ref):
        err = float((a.double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize("N,Cin,Cout,H,W,bias", [(2, 256, 128, 29, 50, True), (1, 512, 256, 13, 40, False)])
def test_conv1x1_training_node_matches_float64(dev, N, Cin, Cout, H, W, bias):
    """train_conv._Conv1x1 (the image FPN's lateral convolutions under autograd: forward / data gradient on the library's 1x1 GEMM, weight
    gradient on srf_conv_wgrad_nhwc) against float64 autograd"""
    from srfdet3d_amd import train_conv
    torch.manual_seed(Cin + Cout)
    conv = torch.nn.Conv2d(Cin, Cout, 1, bias=bias).to(dev)
    x = torch.randn(N, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(N, Cout, H, W, device=dev)
    y = train_conv.conv2d(conv, x)
    assert type(y.grad_fn).__name__ == "_Conv1x1Backward" and y.stride(1) == 1
    y.backward(g)
    got = [y.detach(), x.grad, conv.weight.grad] + ([conv.bias.grad] if bias else [])
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True) if bias else None
    yr = F.conv2d(xd, wd, bd)
    yr.backward(g.double())
    ref = [yr.detach(), xd.grad, wd.grad] + ([bd.grad] if bias else [])
    for name, a, b in zip(["y", "dx", "dW", "db"], got, ref):
        err = float((a.double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err < 2e-5, (name, err)
