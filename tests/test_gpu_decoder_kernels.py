"""Unit parity of the decoder-stage kernels (csrc/decoder.hip) against torch on the same device-independent inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from srfdet3d_amd import ops

pytestmark = pytest.mark.gpu


def _ln(n, dev, g):
    m = torch.nn.LayerNorm(n).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.rand(n, generator=g) + 0.5)
        m.bias.copy_(torch.randn(n, generator=g) * 0.1)
    return m


@pytest.mark.parametrize("M,K,N", [(200, 128, 384), (200, 128, 8192), (200, 6272, 128), (200, 128, 512), (200, 512, 128),
                                   (200, 128, 10), (900, 128, 128), (9800, 256, 128), (37, 128, 128), (100, 256, 256),
                                   (100, 12544, 256)])
def test_linear_variants(dev, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = F.linear(x.double(), w.double(), b.double())
    tol = dict(rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ops.linear(x, w, b).double(), ref, **tol)
    torch.testing.assert_close(ops.linear(x, w, None, relu1=True).double(), F.linear(x.double(), w.double()).relu(), **tol)
    if N <= 1024:
        ln1, ln2 = _ln(N, dev, g), _ln(N, dev, g)
        res = torch.randn(M, N, generator=g).to(dev)
        with torch.no_grad():
            want = ln2.double()(F.relu(ln1.double()(ref)) + res.double())
            ln1.float(), ln2.float()
            got = ops.linear(x, w, b, ln1=ln1, relu1=True, residual=res, ln2=ln2)
            torch.testing.assert_close(got.double(), want, rtol=5e-5, atol=5e-5)
            want2 = F.relu(F.layer_norm(F.linear(x.double(), w.double()), (N,), ln1.weight.double(), ln1.bias.double(), ln1.eps))
            torch.testing.assert_close(ops.linear(x, w, None, ln1=ln1, relu1=True).double(), want2, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("P,E,H", [(200, 128, 8), (900, 128, 8), (100, 256, 8), (33, 128, 8)])
def test_self_attention_matches_torch_mha(dev, P, E, H):
    torch.manual_seed(P)
    mha = torch.nn.MultiheadAttention(E, H).to(dev).eval()
    x = torch.randn(P, 1, E, device=dev)
    with torch.no_grad():
        want = mha(x, x, value=x)[0][:, 0]
        qkv = ops.linear(x[:, 0].contiguous(), mha.in_proj_weight, mha.in_proj_bias)
        att = ops.self_attention(qkv, H)
        got = ops.linear(att, mha.out_proj.weight, mha.out_proj.bias)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("C,D", [(128, 32), (256, 64)])
def test_dynconv_mid_matches_torch(dev, C, D):
    g = torch.Generator().manual_seed(C)
    R, S = 50, 49
    feats = torch.randn(R, S, C, generator=g).to(dev)
    params = (torch.randn(R, 2 * C * D, generator=g) * 0.1).to(dev)
    n1, n2 = _ln(D, dev, g), _ln(C, dev, g)
    with torch.no_grad():
        w1 = params[:, :C * D].view(R, C, D).double()
        w2 = params[:, C * D:].view(R, D, C).double()
        x = F.relu(F.layer_norm(torch.bmm(feats.double(), w1), (D,), n1.weight.double(), n1.bias.double(), n1.eps))
        want = F.relu(F.layer_norm(torch.bmm(x, w2), (C,), n2.weight.double(), n2.bias.double(), n2.eps))
        got = ops.dynconv_mid(feats, params, n1, n2)
    torch.testing.assert_close(got.double(), want, rtol=1e-4, atol=1e-4)
