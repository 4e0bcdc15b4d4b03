"""Dense-side companions (csrc/dense.hip): the fused eval BatchNorm2d + ReLU pass against torch, and the fused
conv -> BN -> ReLU routing of srfdet3d_amd/dense.py against the plain module chain."""
import pytest
import torch
from torch import nn

from srfdet3d_amd import dense, ops

pytestmark = pytest.mark.gpu


def _bn(c, dev, g, eps=1e-3):
    bn = nn.BatchNorm2d(c, eps=eps)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.3)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return bn.to(dev).eval()


@pytest.mark.parametrize("shape", [(6, 128, 58, 100), (1, 256, 184, 184), (2, 7, 29, 50), (3, 5, 1, 1), (1, 64, 33, 7)])
@pytest.mark.parametrize("relu", [True, False])
def test_channel_affine_matches_bn_eval(dev, shape, relu):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(dev)
    bn = _bn(shape[1], dev, g)
    with torch.no_grad():
        want = bn.double()(x.double())
        bn.float()
        if relu:
            want = want.relu()
        scale, shift = dense._fold_bn2d(bn)
        got = ops.channel_affine(x, scale, shift, relu)
        torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5)
        # in place, and into a channel slice of a wider tensor (the concat-buffer case)
        wide = torch.full((shape[0], shape[1] + 9, *shape[2:]), 7.0, device=dev)
        ops.channel_affine(x, scale, shift, relu, out=wide[:, 4:4 + shape[1]])
        assert torch.equal(wide[:, 4:4 + shape[1]], got)
        assert bool((wide[:, :4] == 7.0).all()) and bool((wide[:, 4 + shape[1]:] == 7.0).all())
        y = x.clone()
        assert ops.channel_affine(y, scale, shift, relu, out=y).data_ptr() == y.data_ptr()
        assert torch.equal(y, got)


def test_run_sequential_equals_module_chain(dev):
    g = torch.Generator().manual_seed(5)
    seq = nn.Sequential(nn.Conv2d(8, 16, 3, 1, 1, bias=False), _bn(16, dev, g), nn.ReLU(inplace=True),
                        nn.Conv2d(16, 16, 3, 2, 1, bias=False), _bn(16, dev, g), nn.ReLU(inplace=True),
                        nn.Conv2d(16, 4, 1, bias=False), _bn(4, dev, g)).to(dev).eval()
    x = torch.randn(2, 8, 40, 36, generator=g).to(dev)
    with torch.no_grad():
        want = seq(x.clone())
        got = dense.run_sequential(seq, x.clone())
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # with grad enabled (training / fine-tuning) the modules run as written
    seq.train()
    y = dense.run_sequential(seq, x.clone().requires_grad_(True))
    assert y.requires_grad


def test_channel_affine_gate_and_identity(dev):
    """per-(sample, channel) scale without shift + residual: the eSE gate and the OSA identity add in one pass."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 10, 17, 12, generator=g).to(dev)
    res = torch.randn(3, 10, 17, 12, generator=g).to(dev)
    gate = torch.rand(3, 10, 1, 1, generator=g).to(dev)
    want = x * gate + res
    got = ops.channel_affine(x.clone(), gate.reshape(-1), None, False, residual=res)
    assert torch.equal(got, want)
    assert torch.equal(ops.channel_affine(x.clone(), gate.reshape(-1), None, False), x * gate)


@pytest.mark.parametrize("N,chans,Cout,H,W", [(2, (128, 160, 160), 256, 12, 20), (1, (32,), 128, 2, 2), (3, (256, 64, 64, 64, 64, 64), 512, 29, 52),
                                              (6, (512, 192, 192, 192, 192, 192), 768, 58, 100), (1, (96, 32), 128, 5, 36)])
def test_conv1x1_over_concat_matches_torch(dev, N, chans, Cout, H, W):
    """srf_conv1x1 (concat never built, BN + ReLU epilogue) against torch conv2d on the concatenation, float64."""
    g = torch.Generator().manual_seed(N + Cout + H)
    xs = [torch.randn(N, c, H, W, generator=g).to(dev) for c in chans]
    K = sum(chans)
    conv = nn.Conv2d(K, Cout, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(Cout, K, 1, 1, generator=g) / K ** 0.5)
    conv = conv.to(dev)
    bn = _bn(Cout, dev, g)
    with torch.no_grad():
        want = bn.double()(conv.double()(torch.cat(xs, 1).double())).relu()
        bn.float(), conv.float()
        got = dense.conv1x1_cat_bn_act(conv, bn, True, xs)
        torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5)
        # bias instead of BatchNorm, no activation (an FPN lateral)
        conv_b = nn.Conv2d(K, Cout, 1).to(dev)
        want = conv_b.double()(torch.cat(xs, 1).double())
        conv_b.float()
        got = dense.conv1x1_cat_bn_act(conv_b, None, False, xs)
        torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5)


def test_conv1x1_falls_back_on_unsupported_shapes(dev):
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(2, 24, 7, 5, generator=g).to(dev), torch.randn(2, 40, 7, 5, generator=g).to(dev)]   # C % 32 != 0, HW odd
    conv = nn.Conv2d(64, 128, 1, bias=False).to(dev)
    bn = _bn(128, dev, g)
    with torch.no_grad():
        want = bn(conv(torch.cat(xs, 1))).relu()
        got = dense.conv1x1_cat_bn_act(conv, bn, True, xs)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W,Ht,Wt", [(6, 256, 116, 200, 58, 100), (1, 128, 184, 184, 92, 92), (2, 8, 23, 24, 12, 12), (1, 4, 15, 20, 8, 10)])
def test_upsample_add_equals_interpolate_plus_add(N, C, H, W, Ht, Wt):
    """the FPN top-down step in one pass: bit-identical to lateral + F.interpolate(top, size=..., mode='nearest')"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * W)
    lat = torch.randn(N, C, H, W, generator=g).to(dev)
    top = torch.randn(N, C, Ht, Wt, generator=g).to(dev)
    assert ops.upsample_add_supported(lat, top)
    want = lat + torch.nn.functional.interpolate(top, size=(H, W), mode="nearest")
    assert torch.equal(ops.upsample_add(lat, top), want)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W", [(6, 256, 232, 400), (2, 16, 116, 200), (1, 8, 58, 100), (1, 3, 29, 50), (2, 5, 7, 9), (1, 2, 3, 3),
                                     (1, 2, 4, 6), (1, 1, 2, 2), (1, 4, 33, 18)])
def test_maxpool3s2_ceil_equals_torch(N, C, H, W):
    """VoVNet's stage pooling: equal to nn.MaxPool2d(3, stride=2, ceil_mode=True), clipped windows included"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 131 + W)
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    want = torch.nn.functional.max_pool2d(x, 3, stride=2, ceil_mode=True)
    got = ops.maxpool3s2_ceil(x)
    assert got.shape == want.shape
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C", [(6, 768), (1, 256), (8, 1024), (2, 64)])
def test_ese_gate_matches_torch(N, C):
    """hsigmoid(fc(global average)) of VoVNet's eSE module as one GEMV launch"""
    from srfdet3d_amd.plugin.vovnet import eSEModule
    dev = torch.device("cuda:0")
    torch.manual_seed(C)
    m = eSEModule(C).to(dev).eval()
    x = torch.randn(N, C, 12, 20, device=dev)
    with torch.no_grad():
        mean = x.mean(dim=(2, 3))
        want = (torch.nn.functional.relu6(m.fc(mean.view(N, C, 1, 1).double().float()) + 3.0) / 6.0).view(N, C)
        got = ops.ese_gate(mean, m.fc.weight, m.fc.bias)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
        # the module itself, fused route against its torch definition
        ref = x * want.view(N, C, 1, 1)
        torch.testing.assert_close(m(x.clone()), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(6, 128, 232, 400), (1, 128, 184, 184), (2, 40, 23, 24), (1, 3, 2, 2), (3, 33, 5, 8)])
def test_to_channels_last_equals_torch(shape):
    dev = torch.device("cuda:0")
    x = torch.randn(*shape, device=dev)
    y = ops.to_channels_last(x)
    want = x.contiguous(memory_format=torch.channels_last)
    assert y.stride() == want.stride() and torch.equal(y, want)
    assert ops.to_channels_last(torch.randn(1, 4, 3, 3, device=dev)).is_contiguous(memory_format=torch.channels_last)   # HW % 4 != 0: torch path


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W", [(6, 128, 232, 400), (1, 128, 184, 184), (1, 256, 92, 92), (2, 12, 23, 23), (1, 3, 5, 7), (1, 2, 1, 1),
                                     (1, 4, 2, 9)])
def test_dwconv3x3s2_bn_relu_matches_torch(N, C, H, W):
    """the depthwise stair of the proposal generator (conv 3x3 s2 p1 groups=C + BN(eval) + ReLU) in one kernel"""
    dev = torch.device("cuda:0")
    torch.manual_seed(C * 7 + H)
    conv = nn.Conv2d(C, C, 3, 2, 1, groups=C, bias=False).to(dev)
    bn = nn.BatchNorm2d(C, eps=1e-3).to(dev).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
        x = torch.randn(N, C, H, W, device=dev)
        with torch.backends.cudnn.flags(enabled=False):
            want = torch.relu(bn(conv(x).double().float()))
            ref64 = torch.relu(torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, 2, 1, 1, C),
                                                              bn.running_mean.double(), bn.running_var.double(), bn.weight.double(),
                                                              bn.bias.double(), False, 0.0, bn.eps))
        got = dense.conv_bn_act(conv, bn, True, x)
        assert got.shape == want.shape
        torch.testing.assert_close(got.double(), ref64, rtol=1e-5, atol=1e-5)
