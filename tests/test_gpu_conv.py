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


# ---- 1x1 convolution (GEMM) on channels-last slices ------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,K,Cout", [
    (1, 3, 5, 32, 8),          # a partial pixel tile, Cout < 32
    (2, 9, 14, 64, 300),       # two channel blocks, Cout not a multiple of 32
    (1, 29, 50, 160, 256),     # small-map path (128-pixel tiles)
    (6, 58, 100, 96, 256),     # 256-pixel tiles, partial last tile
])
def test_conv1x1_nhwc_matches_float64(N, H, W, K, Cout):
    g = torch.Generator().manual_seed(K + Cout)
    x = torch.randn(N, H, W, K, generator=g).to(DEV)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk = ops.pack_conv1x1_nhwc_weights(w)
    y = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True)
    ref = ((x.cpu().double() @ w.cpu().double().t()) * scale.cpu().double() + shift.cpu().double()).relu()
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)
    y = ops.conv1x1_nhwc(x, pk, Cout)
    ref = x.cpu().double() @ w.cpu().double().t()
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)


def test_conv1x1_nhwc_is_a_k_ordered_fma_chain():
    """The f32 MFMA accumulates k ascending with one rounding per product: with small-integer data the result is exact,
    and a slice of a wider buffer is read / written in place."""
    g = torch.Generator().manual_seed(3)
    buf = torch.randint(-4, 5, (2, 6, 7, 80), generator=g).float().to(DEV)
    w = torch.randint(-3, 4, (24, 32), generator=g).float().to(DEV)
    dst = torch.full((2, 6, 7, 40), 7.0, device=DEV)
    ops.conv1x1_nhwc(buf[..., 16:48], ops.pack_conv1x1_nhwc_weights(w), 24, out=dst[..., 8:32])
    ref = buf[..., 16:48] @ w.t()
    assert torch.equal(dst[..., 8:32], ref)
    assert torch.all(dst[..., :8] == 7.0) and torch.all(dst[..., 32:] == 7.0)


@pytest.mark.parametrize("N,H,W,K,Cout", [(2, 9, 15, 64, 96), (3, 16, 16, 32, 256), (1, 23, 29, 96, 40), (2, 1, 5, 32, 300)])
def test_conv1x1_nhwc_pooled_mean_of_stored_outputs(N, H, W, K, Cout):
    """Per-image row tiling: the outputs equal the flat kernel's bit for bit, the mean is the mean of what was stored
    (blocks that end mid-tile with their image do not leak rows of the next image), and it repeats bit for bit."""
    g = torch.Generator().manual_seed(N * H + W)
    x = torch.randn(N, H, W, K, generator=g).to(DEV)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk = ops.pack_conv1x1_nhwc_weights(w)
    y0 = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True)
    y, mean = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True, pool=True)
    assert torch.equal(y, y0)
    ref = y.double().mean(dim=(1, 2))
    assert (mean.double() - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1.0)
    y2, mean2 = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True, pool=True)
    assert torch.equal(mean, mean2) and torch.equal(y, y2)


# ---- streaming NHWC layers ---------------------------------------------------------------------------------------------
def test_nhwc_streaming_layers_match_torch():
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 2, 13, 18, 64
    big = torch.randn(N, H, W, C + 32, generator=g).to(DEV)
    x = big[..., 16:16 + C]                      # a channel slice
    xc = x.permute(0, 3, 1, 2)
    # affine: per-channel scale / shift + ReLU; per-sample gate + residual
    sc, sh = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV)
    y = ops.nhwc_affine(x, sc, sh, True)
    assert torch.equal(y, torch.relu(x * sc + sh))
    gate = torch.rand(N, C, generator=g).to(DEV)
    res = torch.randn(N, H, W, C, generator=g).to(DEV)
    y = ops.nhwc_affine(x, gate, None, False, residual=res)
    assert torch.equal(y, x * gate.view(N, 1, 1, C) + res)
    # global average pool (deterministic order differs from torch's: a few ulp)
    m = ops.nhwc_colmean(x)
    assert torch.allclose(m, x.mean(dim=(1, 2)), rtol=1e-5, atol=1e-6)
    for Cw in (768, 1024):
        xx = torch.randn(1, 7, 9, Cw, generator=g).to(DEV)
        assert torch.allclose(ops.nhwc_colmean(xx), xx.mean(dim=(1, 2)), rtol=1e-5, atol=1e-6)
    # max pool 3 / 2 ceil
    y = ops.nhwc_maxpool3s2_ceil(x)
    ref = F.max_pool2d(xc, 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
    assert torch.equal(y, ref)
    # nearest upsample + add
    top = torch.randn(N, 7, 9, C, generator=g).to(DEV)
    lat = x.clone()
    y = ops.nhwc_upsample_add(lat, top)
    ref = x + F.interpolate(top.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(y, ref) and y.data_ptr() == lat.data_ptr()
    # depthwise 3x3 stride 2 + BN + ReLU
    w = torch.randn(C, 1, 3, 3, generator=g).to(DEV)
    y = ops.nhwc_dwconv3x3s2(x, w, sc, sh, True)
    ref = torch.relu(F.conv2d(xc, w, stride=2, padding=1, groups=C) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert (y - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def _randomize_bn(m, g):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)


def test_vovnet_fpn_channels_last_path_matches_module_path(monkeypatch):
    """VoVNet-99 -> FPN -> img_convs on the NHWC kernels against the same modules run through torch (SRF_IMG_NHWC=0):
    features within 2e-4 of the level's max (the bar of the SECOND / FPN tests)."""
    from srfdet3d_amd.compat.necks import FPN
    from srfdet3d_amd.plugin.vovnet import VoVNet
    g = torch.Generator().manual_seed(2)
    torch.manual_seed(2)
    net = VoVNet("V-99-eSE", out_features=["stage2", "stage3", "stage4", "stage5"])
    fpn = FPN([256, 512, 768, 1024], 256, 4, add_extra_convs="on_output", relu_before_extra_convs=True)
    _randomize_bn(net, g)
    net, fpn = net.to(DEV).eval(), fpn.to(DEV).eval()
    x = torch.randn(2, 3, 96, 160, generator=g).to(DEV)
    with torch.no_grad():
        monkeypatch.setenv("SRF_IMG_NHWC", "0")
        ref = fpn(list(net(x).values()))
        monkeypatch.setenv("SRF_IMG_NHWC", "1")
        feats = net(x)
        assert all(f.stride(1) == 1 for f in feats.values())   # channels_last views of the NHWC buffers
        out = fpn(list(feats.values()))
    for o, r in zip(out, ref):
        assert o.shape == r.shape
        assert (o - r).abs().max().item() <= 2e-4 * r.abs().max().item()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(6, 232, 400, 128, 128), (6, 116, 200, 160, 160), (6, 58, 100, 192, 192),
                                            (6, 29, 50, 224, 224), (1, 184, 184, 256, 128)])
def test_winograd_equals_the_direct_convolution_at_full_layer_sizes(N, H, W, Cin, Cout):
    """BASELINE.json's full sizes (six 928 x 1600 cameras -> the 232 x 400 ... 29 x 50 maps of VoVNet-99, the 184 x 184 BEV map):
    the Winograd kernel against an INDEPENDENT kernel of this library, the implicit-im2col GEMM (`srf_conv_gemm_nhwc`, every
    output one k-ordered fma chain, itself pinned to float64 at small sizes above) -- within 2e-5 of the map's maximum, with
    scale / shift / ReLU, on every pixel incl. the borders and the partial tile blocks."""
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    yw = ops.wino3x3(x, ops.pack_wino3x3_weights(w), Cout, scale, shift, True)
    yd = ops.conv_gemm_nhwc(x, ops.pack_conv_gemm_weights(w), Cout, (3, 3), 1, 1, scale, shift, True)
    assert yw.shape == yd.shape
    assert (yw - yd).abs().max().item() <= 2e-5 * yd.abs().max().item()
    assert (yw > 0).float().mean().item() > 0.2          # the comparison is not between two all-zero maps


@pytest.mark.parametrize("N,HW,K,Cout", [(6, 232 * 400, 768, 256), (6, 116 * 200, 1312, 512), (6, 58 * 100, 1728, 768),
                                         (6, 29 * 50, 2144, 1024)])
def test_conv1x1_at_full_layer_sizes_against_rocblas(N, HW, K, Cout):
    """The OSA concat convolutions of an LC frame at their real sizes: outputs within 3e-5 of the map's maximum of torch.mm
    (rocBLAS, another summation order), the fused eSE mean equal to the mean of what was stored, the same bits on a rerun."""
    g = torch.Generator().manual_seed(K)
    x = torch.randn(N, 1, HW, K, generator=g).to(DEV)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk = ops.pack_conv1x1_nhwc_weights(w)
    y, mean = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True, pool=True)
    ref = torch.relu(torch.mm(x.view(N * HW, K), w.t()) * scale + shift).view(N, 1, HW, Cout)
    assert (y - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    m_ref = y.double().mean(dim=(1, 2))
    assert (mean.double() - m_ref).abs().max().item() <= 1e-5 * m_ref.abs().max().item()
    y2, mean2 = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True, pool=True)
    assert torch.equal(y, y2) and torch.equal(mean, mean2)


_WINO_FORMS = r"""
import sys, torch
from srfdet3d_amd import ops
outs = []
for (N, H, W, Cin, Cout) in [(2, 37, 53, 64, 160), (1, 64, 64, 32, 32), (3, 20, 132, 40, 96), (1, 58, 100, 224, 224), (2, 9, 11, 16, 20)]:
    g = torch.Generator().manual_seed(Cout + W)
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    shift = torch.randn(Cout, generator=g).cuda()
    outs.append(ops.wino3x3(x, ops.pack_wino3x3_weights(w), Cout, scale, shift, True).cpu())
torch.save(outs, sys.argv[1])
"""


def test_winograd_half_block_kernel_gives_the_same_bits(tmp_path):
    """A last channel block with <= 32 real channels can run on the half-block kernel (waves split the frequencies instead
    of the channels, the upper half hands its accumulators over through LDS): the output transform then performs the same
    operations in the same order, so the results are bit-identical to the one-launch form, whichever the launcher picks.
    Forms: SRF_WINO_HALF = 0 one-launch / 1 last block as half blocks / 2 every block as two half blocks / 3 the second half of the
    work items as two half blocks each (the form a partly filled last round takes), each on the 8 x 8, 16 x 4 and 32 x 2 tile
    blocks (SRF_WINO_TWL).  The knobs are read once per process: one interpreter per setting (tests/forms.py)."""
    from forms import run_forms
    # every half-block form on the default tile-block shape, every tile-block shape on the one-launch and the all-half forms
    settings = ([{"SRF_WINO_HALF": h} for h in ("0", "1", "2", "3")] + [{"SRF_WINO_HALF": h, "SRF_WINO_TWL": t} for h in ("0", "2") for t in ("1", "2", "3")])
    res = run_forms(_WINO_FORMS, settings, tmp_path)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)
    # and the default choice against float64
    N, H, W, Cin, Cout = 2, 37, 53, 64, 160
    g = torch.Generator().manual_seed(Cout + W)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1)
    ref = (ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).relu().permute(0, 2, 3, 1)
    assert (res[0][0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


_GEMM_FORMS = r"""
import sys, torch
from srfdet3d_amd import ops
outs = []
for (N, HW, K, Cout, mode) in [(1, 128 * 404, 64, 256, "flat"), (2, 128 * 200 + 40, 64, 256, "pool"), (3, 128 * 86 + 1, 32, 384, "pool"),
                               (1, 128 * 260 + 77, 96, 400, "flat"), (2, 100 * 257, 32, 256, "top")]:
    g = torch.Generator().manual_seed(HW + Cout)
    x = torch.randn(N, 1, HW, K, generator=g).cuda()
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).cuda()
    shift = torch.randn(Cout, generator=g).cuda()
    pk = ops.pack_conv1x1_nhwc_weights(w)
    if mode == "pool":
        y, m = ops.conv1x1_nhwc(x, pk, Cout, None, shift, True, pool=True)
        outs.append((y.cpu(), m.cpu()))
    elif mode == "top":
        top = torch.randn(N, 50, 129, Cout, generator=torch.Generator().manual_seed(5)).cuda()
        outs.append((ops.conv1x1_nhwc(x.view(N, 100, 257, K), pk, Cout, None, shift, False, top=top).cpu(),))
    else:
        outs.append((ops.conv1x1_nhwc(x, pk, Cout, None, shift, True).cpu(),))
    if mode != "top":
        ref = torch.relu(torch.mm(x.view(N * HW, K), w.t()) + shift).view(N, 1, HW, Cout)
        assert (outs[-1][0].cuda() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
torch.save(outs, sys.argv[1])
"""


def test_conv1x1_mixed_tiles_give_the_same_bits(tmp_path):
    """A partly filled last round of 128 x 128 tiles runs as 64 x 64 tiles in the same launch (srf_conv1x1_nhwc_mixed_k,
    SRF_GEMM_TAIL=1, the default) or as one more round of big tiles (SRF_GEMM_TAIL=0): every output is the same k-ordered fma
    chain, so the map is bit-identical; the pooled means add the blocks of an image in another grouping and agree to rounding."""
    from forms import run_forms
    res = run_forms(_GEMM_FORMS, [{"SRF_GEMM_TAIL": "0"}, {"SRF_GEMM_TAIL": "1"}], tmp_path)
    for a, b in zip(*res):
        assert torch.equal(a[0], b[0])
        if len(a) > 1:
            m_ref = a[0].double().mean(dim=(1, 2))
            for o in (a, b):
                assert (o[1].double() - m_ref).abs().max().item() <= 1e-5 * m_ref.abs().max().item()


@pytest.mark.parametrize("N,H,W,Ht,Wt,K,Cout", [(2, 12, 20, 6, 10, 64, 96), (3, 29, 50, 15, 25, 128, 256), (1, 7, 9, 4, 5, 32, 40),
                                                   (6, 116, 200, 58, 100, 512, 256)])
def test_conv1x1_topdown_equals_conv_then_upsample_add(N, H, W, Ht, Wt, K, Cout):
    """The FPN top-down step as the epilogue of the lateral convolution adds the same two floats as the separate pass:
    identical bits (odd sizes: nearest upsampling by size, floor(py Ht / H))."""
    g = torch.Generator().manual_seed(H * W + K)
    x = torch.randn(N, H, W, K, generator=g).to(DEV)
    top = torch.randn(N, Ht, Wt, Cout, generator=g).to(DEV)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk = ops.pack_conv1x1_nhwc_weights(w)
    for relu in (False, True):
        want = ops.conv1x1_nhwc(x, pk, Cout, None, shift, relu)
        ops.nhwc_upsample_add(want, top)
        got = ops.conv1x1_nhwc(x, pk, Cout, None, shift, relu, top=top)
        assert torch.equal(got, want)
    ref = F.interpolate(top.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)
    plain = ops.conv1x1_nhwc(x, pk, Cout, None, shift, False)
    assert torch.equal(ops.conv1x1_nhwc(x, pk, Cout, None, shift, False, top=top), plain + ref)


def test_vovnet_training_runs_the_frozen_prefix_on_the_inference_kernels(monkeypatch):
    """Config 4 trains with `frozen_stages=2, norm_eval=True` (configs/nus/srfdet_voxel_nusc_LC.py:44-54): stem, stage2 and
    stage3 carry no gradient.  With autograd recording they run on the channels-last inference kernels under no_grad; the
    stage outputs equal the all-module path within 2e-4 of the level's max and the trainable stages still get gradients (their
    3x3 layers through train_conv._Wino43Conv, compared here with the torch / MIOpen module path of SRF_IMG_NHWC=0)."""
    from srfdet3d_amd.plugin.vovnet import VoVNet
    g = torch.Generator().manual_seed(4)
    torch.manual_seed(4)
    net = VoVNet("V-99-eSE", out_features=["stage2", "stage3", "stage4", "stage5"], frozen_stages=2, norm_eval=True)
    _randomize_bn(net, g)
    net = net.to(DEV).train()
    assert not net.stage3.training and not any(p.requires_grad for p in net.stage3.parameters())
    assert any(p.requires_grad for p in net.stage4.parameters())
    x = torch.randn(2, 3, 96, 160, generator=g).to(DEV)
    monkeypatch.setenv("SRF_IMG_NHWC", "0")
    ref = net(x)
    monkeypatch.setenv("SRF_IMG_NHWC", "1")
    out = net(x)
    for k in ref:
        # round 3: the trainable remainder runs channels-last (train_conv.py), the frozen prefix hands its maps over without a copy
        assert out[k].shape == ref[k].shape and (out[k].is_contiguous() or out[k].stride(1) == 1)
        assert (out[k] - ref[k]).abs().max().item() <= 2e-4 * ref[k].abs().max().item(), k
    assert not out["stage3"].requires_grad and out["stage4"].requires_grad
    out["stage5"].square().mean().backward()
    gr = next(p for p in net.stage4.parameters() if p.requires_grad).grad
    assert gr is not None and torch.isfinite(gr).all() and gr.abs().max() > 0
    assert all(p.grad is None for p in net.stage2.parameters())


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad", [(2, 13, 18, 32, 48, 3, 2, 1), (1, 184, 184, 128, 256, 3, 2, 1),
                                                         (6, 32, 48, 64, 128, 3, 2, 1), (1, 23, 23, 128, 128, 3, 2, 1),
                                                         (1, 9, 11, 64, 40, 3, 1, 1)])
def test_conv_gemm_nhwc_matches_torch(N, H, W, Cin, Cout, k, stride, pad):
    """Implicit-im2col GEMM for the strided 3x3 layers; float64 CPU convolution as the reference; run twice: bitwise equal
    (no atomics)."""
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk = ops.pack_conv_gemm_weights(w)
    y = ops.conv_gemm_nhwc(x, pk, Cout, (k, k), stride, pad, scale, shift, True)
    y2 = ops.conv_gemm_nhwc(x, pk, Cout, (k, k), stride, pad, scale, shift, True)
    assert torch.equal(y, y2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.cpu().double(), stride=stride, padding=pad)
    ref = (ref * scale.cpu().double().view(1, -1, 1, 1) + shift.cpu().double().view(1, -1, 1, 1)).relu().permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("N,Cin,H,W", [(3, 3, 37, 50), (2, 3, 64, 300), (1, 4, 9, 131), (2, 1, 8, 8), (1, 3, 129, 257)])
def test_stem_conv_nchw_matches_torch(N, Cin, H, W):
    """Odd and even sizes, maps wider and narrower than the 4 x 64-pixel workgroup tile, Cin = 1 .. 4 (k padded to even)."""
    g = torch.Generator().manual_seed(4 + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(64, Cin, 3, 3, generator=g) / 5).to(DEV)
    scale = (torch.rand(64, generator=g) + 0.5).to(DEV)
    shift = torch.randn(64, generator=g).to(DEV)
    y = ops.stem_conv_nchw(x, w, scale, shift, True)
    ref = (F.conv2d(x.cpu().double(), w.cpu().double(), stride=2, padding=1) * scale.cpu().double().view(1, -1, 1, 1)
           + shift.cpu().double().view(1, -1, 1, 1)).relu().permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert (y.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


# ---- the LDS-free GEMM (csrc/gemm_direct.hip) ------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,K,Cout,ld_in,ld_out", [
    (1, 3, 5, 32, 8, 32, 8),             # a partial row tile, Cout < 32
    (2, 9, 14, 64, 300, 96, 320),        # three column tiles (the last one partial), slices of wider buffers
    (3, 29, 50, 160, 256, 160, 256),     # rows of several images, partial last block per image in the pooled form
    (2, 58, 100, 96, 128, 128, 128),
])
def test_conv1x1_direct_gives_the_bits_of_the_lds_kernel(monkeypatch, N, H, W, K, Cout, ld_in, ld_out):
    """`srf_conv1x1_nhwc_direct*` against `srf_conv1x1_nhwc*`: every output is the same fma chain, so plain, top-down and pooled
    outputs must be bitwise equal; the pooled mean (block sums added in another fixed order) within 1e-6 relative."""
    g = torch.Generator().manual_seed(K + Cout + H)
    xb = torch.randn(N, H, W, ld_in, generator=g).to(DEV)
    x = xb[..., ld_in - K:]
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    top = torch.randn(N, (H + 1) // 2, (W + 1) // 2, Cout, generator=g).to(DEV)
    pk, pd = ops.pack_conv1x1_nhwc_weights(w), ops.pack_conv1x1_nhwc_direct_weights(w)
    res = {}
    for force in ("0", "1"):
        monkeypatch.setenv("SRF_GEMM_DIRECT", force)
        ob = torch.full((N, H, W, ld_out), 7.0, device=DEV)
        y = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True, out=ob[..., :Cout], packed_direct=pd)
        assert torch.all(ob[..., Cout:] == 7.0)                   # nothing written outside the slice
        yt = ops.conv1x1_nhwc(x, pk, Cout, None, shift, False, top=top, packed_direct=pd)
        yp, mean = ops.conv1x1_nhwc(x, pk, Cout, scale, shift, True, pool=True, packed_direct=pd)
        res[force] = (y.clone(), yt, yp, mean)
    for a, b in zip(res["0"][:3], res["1"][:3]):
        assert torch.equal(a, b)
    torch.testing.assert_close(res["0"][3], res["1"][3], rtol=1e-6, atol=1e-6)
    ref = _ref(x, w.view(Cout, K, 1, 1), scale, shift, True)
    assert (res["1"][0].cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    torch.testing.assert_close(res["1"][3], res["1"][2].mean((1, 2)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,HW,K,Cout", [(6, 232 * 400, 768, 256), (6, 58 * 100, 1728, 768)])
def test_conv1x1_direct_at_full_layer_sizes(monkeypatch, N, HW, K, Cout):
    """BASELINE.json's full sizes (the stage-2 and stage-4 concat convolutions of VoVNet-99 on six 928 x 1600 views): the
    launch-size rule picks the LDS-free kernel by itself; bits equal to the LDS kernel's on every pixel."""
    g = torch.Generator().manual_seed(K)
    x = torch.randn(N, 1, HW, K, generator=g).relu().to(DEV)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    pk, pd = ops.pack_conv1x1_nhwc_weights(w), ops.pack_conv1x1_nhwc_direct_weights(w)
    assert ops.conv1x1_direct_wanted(N * HW, Cout)
    yd, md = ops.conv1x1_nhwc(x, pk, Cout, None, shift, True, pool=True, packed_direct=pd)
    monkeypatch.setenv("SRF_GEMM_DIRECT", "0")
    yl, ml = ops.conv1x1_nhwc(x, pk, Cout, None, shift, True, pool=True, packed_direct=pd)
    assert torch.equal(yd, yl)
    torch.testing.assert_close(md, ml, rtol=1e-6, atol=1e-6)
