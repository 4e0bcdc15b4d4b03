"""srf_conv1x1_nhwc_split (csrc/gemm_split.hip): the 1x1 convolutions (vovnet.py:222-223 `concat` layers, FPN laterals) as an
f32 GEMM on the bf16 MFMA through an EXACT three-way split of both operands.

What is tested, and what the claim "f32-accurate" means here:
* layout / plane bookkeeping on data where the answer is exact: operands whose significands need one, two and three bf16 planes;
* against float64: the error of the split kernel is held to the error bound of an f32 fma chain (gamma ~ K * 2^-24 * sum |a b| is
  the textbook bound; the f32 MFMA chain measures 1.7-3.0e-7 of sum |a b|, this kernel 2.0-2.8e-7) and, on the same data, to
  twice the error the f32-MFMA kernel (`srf_conv1x1_nhwc`) actually commits -- tolerances written below;
* the three epilogues (plain, pooled, top-down), channel slices, row remainders, every column-tile count used by the model;
* deterministic (bitwise repeatable)."""
import numpy as np
import pytest
import torch

from srfdet3d_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _always_split(monkeypatch):
    """ops.conv1x1_nhwc takes the split kernel only for launches of >= ops.GEMM_SPLIT_MIN_TILES (128) tiles; these tests exercise it on small shapes too"""
    monkeypatch.setenv("SRF_GEMM_SPLIT_MIN", "0")


def _split(x, w, **kw):
    Cout = w.shape[0]
    return ops.conv1x1_nhwc(x, None, Cout, packed_split=ops.pack_conv1x1_nhwc_split_weights(w), **kw)


def _chain(x, w, **kw):
    Cout = w.shape[0]
    return ops.conv1x1_nhwc(x, ops.pack_conv1x1_nhwc_weights(w), Cout, **kw)


@pytest.mark.parametrize("bits_x,bits_w,K", [(7, 7, 64), (17, 1, 32), (1, 17, 32), (10, 10, 32), (22, 1, 32)])
def test_split_is_exact_where_the_answer_is_exact(dev, bits_x, bits_w, K):
    """Integers of `bits` significant bits need ceil(bits / 8) bf16 planes; with few non-zero terms per output every partial
    product and every partial sum is an integer below 2^24, so ANY correct summation order gives the exact integer.  (17, 1):
    x = xh + xm + xl against w = wh: the hl / lh products; (10, 10): hh, hm, mh, mm; (22, 1): three full planes.)"""
    g = torch.Generator().manual_seed(bits_x * 100 + bits_w)
    N, H, W, Cout = 2, 9, 13, 200     # 234 rows: a partial row block; 200 channels: a partial column tile
    nz = 2 if bits_x + bits_w > 18 else 6     # non-zero terms per row of x
    x = torch.zeros(N * H * W, K, dtype=torch.int64)
    for r in range(x.shape[0]):
        cols = torch.randperm(K, generator=g)[:nz]
        x[r, cols] = torch.randint(-(1 << bits_x) + 1, 1 << bits_x, (nz,), generator=g)
    w = torch.randint(-(1 << bits_w) + 1, 1 << bits_w, (Cout, K), generator=g)
    want = (x @ w.t()).view(N, H, W, Cout)
    assert want.abs().max() < (1 << 24)
    got = _split(x.float().view(N, H, W, K).to(dev), w.float().to(dev))
    assert torch.equal(got.cpu().double(), want.double())


@pytest.mark.parametrize("N,H,W,K,Cout", [(1, 31, 33, 96, 100), (2, 40, 50, 768, 256), (1, 29, 50, 2144, 1024), (3, 20, 20, 1312, 512)])
def test_split_matches_float64_like_the_f32_chain(dev, N, H, W, K, Cout):
    g = torch.Generator().manual_seed(K + Cout)
    x = torch.relu(torch.randn(N, H, W, K, generator=g) * 1.5 + 0.2)          # post-ReLU activations: ~45 % zeros
    w = torch.randn(Cout, K, generator=g) / K ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    xd, wd = x.to(dev), w.to(dev)
    got = _split(xd, wd).cpu().double()
    chain = _chain(xd, wd).cpu().double()
    x2 = x.view(-1, K).double()
    ref = x2 @ w.double().t()
    mag = x2.abs() @ w.double().abs().t()                                       # sum |a b| per output
    e_split = ((got.view(-1, Cout) - ref).abs() / mag.clamp_min(1e-30)).max().item()
    e_chain = ((chain.view(-1, Cout) - ref).abs() / mag.clamp_min(1e-30)).max().item()
    # the f32 chain's measured error grows like sqrt(K) * 2^-24; 6e-7 of sum |a b| covers K <= 4096 for either kernel
    assert e_split <= 6e-7, (e_split, e_chain)
    assert e_split <= 2.0 * e_chain + 1e-7, (e_split, e_chain)
    # epilogue: scale / shift / ReLU on the accumulator, as the f32 kernels
    got2 = _split(xd, wd, scale=scale.to(dev), shift=shift.to(dev), relu=True).cpu().double()
    want2 = torch.relu(ref * scale.double() + shift.double()).view(N, H, W, Cout)
    assert (got2 - want2).abs().max().item() <= 2e-6 * max(1.0, want2.abs().max().item())
    # deterministic
    assert torch.equal(_split(xd, wd).cpu().double(), got)


def test_split_reads_and_writes_channel_slices(dev):
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(2, 17, 19, 160, generator=g).to(dev)
    dst = torch.full((2, 17, 19, 72), 7.0, device=dev)
    w = (torch.randn(24, 64, generator=g) / 8).to(dev)
    _split(buf[..., 32:96], w, out=dst[..., 8:32])
    want = buf[..., 32:96].double().cpu() @ w.double().cpu().t()
    assert (dst[..., 8:32].double().cpu() - want).abs().max().item() < 1e-5
    assert torch.all(dst[..., :8] == 7.0) and torch.all(dst[..., 32:] == 7.0)


@pytest.mark.parametrize("N,H,W,K,Cout", [(3, 23, 27, 160, 96), (2, 58, 100, 1728, 768)])
def test_split_pooled_mean_of_stored_outputs(dev, N, H, W, K, Cout):
    g = torch.Generator().manual_seed(11)
    x = torch.relu(torch.randn(N, H, W, K, generator=g)).to(dev)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(dev)
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
    y0 = _split(x, w, scale=scale, shift=shift, relu=True)
    y, mean = _split(x, w, scale=scale, shift=shift, relu=True, pool=True)
    assert torch.equal(y, y0)                                               # the pooled form stores the same outputs
    want = y.double().mean(dim=(1, 2))
    assert (mean.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    y2, mean2 = _split(x, w, scale=scale, shift=shift, relu=True, pool=True)
    assert torch.equal(mean2, mean)                                         # fixed summation order
    yc, meanc = _chain(x, w, scale=scale, shift=shift, relu=True, pool=True)
    assert (mean - meanc).abs().max().item() <= 2e-6 * max(1.0, meanc.abs().max().item())


@pytest.mark.parametrize("N,H,W,Ht,Wt,K,Cout", [(2, 20, 30, 10, 15, 64, 128), (1, 29, 50, 15, 25, 768, 256), (2, 13, 21, 7, 11, 96, 40)])
def test_split_topdown_equals_conv_then_upsample_add(dev, N, H, W, Ht, Wt, K, Cout):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, W, K, generator=g).to(dev)
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(dev)
    shift = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    top = torch.randn(N, Ht, Wt, Cout, generator=g).to(dev)
    plain = _split(x, w, shift=shift)
    up = torch.nn.functional.interpolate(top.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)
    got = _split(x, w, shift=shift, top=top)
    assert torch.equal(got, plain + up)                                      # the same float is added once, as in the f32 kernels


def test_camera_executor_routes_1x1_layers_to_the_split_kernel_by_default(dev, monkeypatch):
    """nhwc.conv1x1 (OSA `concat` layers, FPN laterals) takes the split kernel unless SRF_GEMM_SPLIT=0, and both routes agree
    within the f32 chain's own error."""
    from srfdet3d_amd import nhwc
    g = torch.Generator().manual_seed(9)
    conv = torch.nn.Conv2d(256, 128, 1, bias=True).to(dev)
    x = torch.relu(torch.randn(2, 24, 40, 256, generator=g)).to(dev)
    with torch.no_grad():
        monkeypatch.delenv("SRF_GEMM_SPLIT", raising=False)
        a = nhwc.conv1x1(x, conv).clone()
        assert hasattr(conv, "_srf_gemm_split") and not hasattr(conv, "_srf_gemm")
        monkeypatch.setenv("SRF_GEMM_SPLIT", "0")
        b = nhwc.conv1x1(x, conv).clone()
        assert hasattr(conv, "_srf_gemm") or hasattr(conv, "_srf_gemm_direct")
        want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu()).permute(0, 2, 3, 1)
    assert (a.double().cpu() - want).abs().max().item() <= 3e-6 * want.abs().max().item()
    assert (b.double().cpu() - want).abs().max().item() <= 3e-6 * want.abs().max().item()
    # with the default threshold a launch this small (15 tiles) stays on the f32-MFMA kernels (ops.gemm_split_wanted)
    monkeypatch.delenv("SRF_GEMM_SPLIT", raising=False)
    monkeypatch.delenv("SRF_GEMM_SPLIT_MIN", raising=False)
    conv2 = torch.nn.Conv2d(256, 128, 1, bias=True).to(dev)
    with torch.no_grad():
        nhwc.conv1x1(x, conv2)
    assert not hasattr(conv2, "_srf_gemm_split") and (hasattr(conv2, "_srf_gemm") or hasattr(conv2, "_srf_gemm_direct"))
    assert ops.gemm_split_wanted(6 * 232 * 400, 256) and not ops.gemm_split_wanted(92 * 92, 128)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad", [(2, 37, 41, 64, 128, 3, 2, 1), (1, 46, 46, 128, 128, 3, 2, 1), (2, 20, 24, 32, 40, 3, 1, 1),
                                                       (1, 33, 29, 64, 96, 1, 2, 0)])
def test_conv_gemm_split_matches_float64(dev, N, H, W, Cin, Cout, k, stride, pad):
    """srf_conv_gemm_nhwc_split (implicit im2col on the split GEMM: VoVNet stem_3, the stride-2 layers of SECOND / BEV FPN) against
    torch's conv2d in float64, and against the f32-MFMA form on the same data."""
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.relu(torch.randn(N, H, W, Cin, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    xd, wd = x.to(dev), w.to(dev)
    got = ops.conv_gemm_nhwc(xd, None, Cout, (k, k), stride, pad, scale.to(dev), shift.to(dev), True,
                             packed_split=ops.pack_conv_gemm_split_weights(wd)).cpu().double()
    f32 = ops.conv_gemm_nhwc(xd, ops.pack_conv_gemm_weights(wd), Cout, (k, k), stride, pad, scale.to(dev), shift.to(dev), True).cpu().double()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=stride, padding=pad)
    ref = torch.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    tol = 2e-6 * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol
    assert (got - ref).abs().max().item() <= 2.0 * (f32 - ref).abs().max().item() + 0.25 * tol
