"""srf_conv_wgrad_nhwc (csrc/wgrad.hip): the weight gradient of the trainable stride-1 convolutions of config 4 (tools/train.py:220-234;
vovnet.py:354-374, the image FPN, srfdet_head.py:404-416) as an f32 GEMM over the pixels on the bf16 MFMA.

Checked against torch's own autograd in float64 (conv2d -> backward), on the layer shapes of the configuration and on shapes that
exercise every edge of the kernel: channel counts that are not multiples of 128 (partial row / column tiles), pixel counts that are not
multiples of 32 (a partial last chunk), several pixel ranges (the fixed-order reduction), image borders (the zero padding of the 3x3
taps), channel SLICES of wider buffers (pixel pitch != channels).  Tolerance: the f32 chain's own, relative to sum |g x| per output."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from srfdet3d_amd import ops

pytestmark = pytest.mark.gpu


def _ref(g, x, k):
    """float64 dW and the magnitude sum |g| * |x| per output through torch's autograd"""
    N, H, W, Cin = x.shape
    Cout = g.shape[3]
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w, padding=k // 2)
    y.backward(g.permute(0, 3, 1, 2).double())
    w2 = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    y2 = F.conv2d(x.permute(0, 3, 1, 2).double().abs(), w2, padding=k // 2)
    y2.backward(g.permute(0, 3, 1, 2).double().abs())
    return w.grad, w2.grad


@pytest.mark.parametrize("N,H,W,Cin,Cout,k", [(2, 9, 40, 64, 96, 3), (1, 13, 33, 160, 224, 3), (3, 7, 50, 192, 192, 1), (1, 29, 50, 224, 224, 3),
                                             (2, 5, 37, 36, 20, 3), (1, 40, 64, 256, 256, 3), (2, 11, 32, 128, 128, 1),
                                             # column tiles over the flattened (tap, channel) axis (Cin % 32 == 0, not a multiple of 128)
                                             (2, 12, 45, 192, 192, 3), (1, 9, 36, 96, 136, 3), (1, 8, 33, 32, 64, 3), (2, 7, 40, 160, 64, 3)])
def test_wgrad_matches_float64_autograd(dev, N, H, W, Cin, Cout, k):
    g_ = torch.Generator().manual_seed(N * 1000 + Cin + Cout + k)
    x = torch.relu(torch.randn(N, H, W, Cin, generator=g_) + 0.2)
    g = torch.randn(N, H, W, Cout, generator=g_) * 0.1
    got = ops.conv_wgrad_nhwc(g.to(dev), x.to(dev), k).cpu().double()
    ref, mag = _ref(g, x, k)
    assert got.shape == ref.shape
    err = ((got - ref).abs() / mag.clamp_min(1e-30)).max().item()
    assert err <= 1e-6, err
    # bitwise repeatable
    again = ops.conv_wgrad_nhwc(g.to(dev), x.to(dev), k).cpu().double()
    assert torch.equal(again, got)


def test_wgrad_reads_channel_slices_and_exact_on_integer_data(dev):
    """operands that are channel slices of wider channels-last buffers; small-integer data: every product and every partial sum is an
    integer below 2^24, so any correct summation gives the exact answer -- checks taps, borders and tile bookkeeping bit for bit"""
    g_ = torch.Generator().manual_seed(7)
    N, H, W = 2, 6, 35
    xb = torch.randint(-3, 4, (N, H, W, 200), generator=g_).float()
    gb = torch.randint(-2, 3, (N, H, W, 144), generator=g_).float()
    x, g = xb[..., 8:8 + 132], gb[..., 4:4 + 68]
    xd, gd = xb.to(dev)[..., 8:8 + 132], gb.to(dev)[..., 4:4 + 68]
    for k in (1, 3):
        got = ops.conv_wgrad_nhwc(gd, xd, k).cpu().double()
        ref, _ = _ref(g, x, k)
        assert torch.equal(got, ref), (got - ref).abs().max().item()


def test_wgrad_of_the_configuration_layers_against_the_library_path(dev):
    """the finest image FPN level at reduced batch (1 x 232 x 400, 256 -> 256, 3x3): several pixel ranges per tile; against MIOpen's
    weight gradient (aten.convolution_backward) within the f32 tolerance of either"""
    g_ = torch.Generator().manual_seed(9)
    N, H, W, C = 1, 116, 200, 256
    x = torch.relu(torch.randn(N, H, W, C, generator=g_)).to(dev)
    g = (torch.randn(N, H, W, C, generator=g_) * 0.05).to(dev)
    got = ops.conv_wgrad_nhwc(g, x, 3)
    w = torch.zeros(C, C, 3, 3, device=dev)
    lib = torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                              (False, True, False))[1]
    scale = lib.abs().max().item()
    assert (got - lib).abs().max().item() <= 2e-5 * scale
