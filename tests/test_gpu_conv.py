"""Channels-last dense convolutions on the f32 MFMA (csrc/conv.hip) against torch's convolution.

Bar: 2e-4 of the output map's max (the bar of the SECOND / FPN feature tests): Winograd F(2x2, 3x3) in f32 differs from a
direct convolution by a few roundings per output.  Small cases are also checked against a float64 CPU convolution."""
import pytest
import torch
import torch.nn.functional as F

from srfdet3d_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x_nhwc, w, scale, shift, relu, dtype=torch.float64):
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2).to("cpu", dtype), w.to("cpu", dtype), padding=w.shape[-1] // 2)
    if scale is not None:
        y = y * scale.to("cpu", dtype).view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.to("cpu", dtype).view(1, -1, 1, 1)
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [
    (1, 2, 2, 8, 8),        # a single tile
    (1, 5, 7, 8, 3),        # odd sizes: clipped last tile row / column, Cout < 32
    (2, 16, 16, 16, 64),    # exactly one block per image... and images stacked inside a block
    (3, 9, 21, 24, 70),     # tile rows of different images inside one block, Cout not a multiple of 64
    (1, 29, 50, 40, 96),    # VoVNet stage 5 map size
    (2, 18, 34, 64, 128),
])
def test_wino3x3_matches_float64(N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cin)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk = ops.pack_wino3x3_weights(w)
    for relu in (False, True):
        y = ops.wino3x3(x, pk, Cout, scale, shift, relu)
        ref = _ref(x, w, scale, shift, relu)
        err = (y.cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0), (err, ref.abs().max().item())
    # no affine at all (FPN: bias only -> shift only; here neither)
    y = ops.wino3x3(x, pk, Cout)
    ref = _ref(x, w, None, None, False)
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)


def test_wino3x3_reads_and_writes_channel_slices():
    """Source and destination are slices of wider NHWC buffers (the OSA concat buffer of VoVNet): only the slice is
    written, neighbours keep their contents."""
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 12, 20
    buf = torch.randn(N, H, W, 96, generator=g).to(DEV)
    before = buf.clone()
    w = (torch.randn(32, 32, 3, 3, generator=g) / 17).to(DEV)
    shift = torch.randn(32, generator=g).to(DEV)
    src, dst = buf[..., 32:64], buf[..., 64:96]
    ops.wino3x3(src, ops.pack_wino3x3_weights(w), 32, None, shift, True, out=dst)
    ref = _ref(before[..., 32:64], w, None, shift, True)
    assert torch.equal(buf[..., :64], before[..., :64])
    assert (buf[..., 64:].cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wino3x3_identity_kernel_is_exact_on_integers():
    """Centre tap = 1: the output equals the input.  With small-integer data every Winograd intermediate is exact."""
    x = torch.randint(-8, 9, (1, 10, 14, 8), generator=torch.Generator().manual_seed(1)).float().to(DEV)
    w = torch.zeros(8, 8, 3, 3)
    for c in range(8):
        w[c, c, 1, 1] = 1.0
    y = ops.wino3x3(x, ops.pack_wino3x3_weights(w.to(DEV)), 8)
    assert torch.equal(y, x)
    # an asymmetric kernel (one off-centre tap) catches transposed / mirrored tiles
    w = torch.zeros(8, 8, 3, 3)
    for c in range(8):
        w[c, (c + 1) % 8, 0, 2] = 1.0   # y[oy][ox][c] = x[oy - 1][ox + 1][c + 1]
    y = ops.wino3x3(x, ops.pack_wino3x3_weights(w.to(DEV)), 8)
    ref = torch.zeros_like(x)
    ref[:, 1:, :-1, :] = x[:, :-1, 1:, :].roll(-1, dims=3)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(6, 58, 100, 192, 192), (1, 184, 184, 128, 128), (2, 116, 200, 160, 160)])
def test_wino3x3_vs_torch_gpu_at_layer_sizes(N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    y = ops.wino3x3(x, ops.pack_wino3x3_weights(w), Cout, scale, shift, True)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = ref.relu().permute(0, 2, 3, 1)
    assert (y - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
