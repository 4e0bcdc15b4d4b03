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
                                   (100, 12544, 256),
                                   (1, 532, 1024), (1, 1024, 800), (2, 1060, 3600), (8, 132, 64), (5, 4, 3)])   # M <= 8: the GEMV path
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


def test_self_attention_batched_equals_the_per_sample_launches(dev):
    """bs > 1 (config 4 trains with two frames per GPU, test-time batches): one launch for the batch, sample by sample the bits
    of the single-sample launch, and the batched nn.MultiheadAttention of the reference (srfdet_head.py:1489-1492)."""
    torch.manual_seed(7)
    bs, P, E, H = 3, 117, 128, 8
    mha = torch.nn.MultiheadAttention(E, H).to(dev).eval()
    x = torch.randn(P, bs, E, device=dev)
    with torch.no_grad():
        want = mha(x, x, value=x)[0]                                     # (P, bs, E)
        rows = x.permute(1, 0, 2).reshape(bs * P, E).contiguous()        # sample-major, as the head lays its proposals out
        qkv = ops.linear(rows, mha.in_proj_weight, mha.in_proj_bias)
        att = ops.self_attention(qkv, H, batch=bs)
        for b in range(bs):
            assert torch.equal(att[b * P:(b + 1) * P], ops.self_attention(qkv[b * P:(b + 1) * P].contiguous(), H))
        got = ops.linear(att, mha.out_proj.weight, mha.out_proj.bias).view(bs, P, E).permute(1, 0, 2)
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


@pytest.mark.parametrize("R,F_,n_cls,n_reg,ncls,Dd", [(200, 512, 2, 3, 10, 10), (900, 512, 2, 3, 10, 10), (37, 256, 1, 1, 3, 8),
                                                      (64, 128, 4, 4, 32, 10), (1, 512, 0, 0, 10, 10)])
def test_stage_tail_matches_per_op_chain(dev, R, F_, n_cls, n_reg, ncls, Dd):
    """srf_stage_tail (FFN + norm3 + towers + logits + deltas + apply_deltas in one launch) against the same chain on
    srf_linear / srf_apply_deltas, and against torch float64 (srfdet_head.py:1506-1520, :1534-1625)."""
    C = 128
    g = torch.Generator().manual_seed(R + F_ + n_cls)
    nn = torch.nn

    def lin(i, o, bias=True):
        m = nn.Linear(i, o, bias=bias)
        with torch.no_grad():
            m.weight.copy_(torch.randn(o, i, generator=g) / i ** 0.5)
            if bias:
                m.bias.copy_(torch.randn(o, generator=g) * 0.1)
        return m.to(dev)

    lin1, lin2, norm3 = lin(C, F_), lin(F_, C), _ln(C, dev, g)
    cls_layers = [(lin(C, C, False), _ln(C, dev, g)) for _ in range(n_cls)]
    reg_layers = [(lin(C, C, False), _ln(C, dev, g)) for _ in range(n_reg)]
    logits_fc, deltas_fc = lin(C, ncls), lin(C, Dd)
    obj = torch.randn(R, C, generator=g).to(dev)
    boxes = torch.cat([torch.rand(R, 3, generator=g) * 100 - 50, torch.randn(R, 3, generator=g) * 0.3 + 0.5,
                       torch.randn(R, Dd - 6, generator=g)], 1).to(dev)
    w6, rng, clamp = [2.0, 2.0, 2.0, 1.0, 1.0, 1.0], [-55.2, -55.2, -5.0, 55.2, 55.2, 3.0], 5.0
    with torch.no_grad():
        hid = ops.linear(obj, lin1.weight, lin1.bias, relu1=True)
        o2 = ops.linear(hid, lin2.weight, lin2.bias, residual=obj, ln2=norm3)
        cf, rf = o2, o2
        for m, n in cls_layers:
            cf = ops.linear(cf, m.weight, None, ln1=n, relu1=True)
        for m, n in reg_layers:
            rf = ops.linear(rf, m.weight, None, ln1=n, relu1=True)
        lg = ops.linear(cf, logits_fc.weight, logits_fc.bias)
        pr = ops.apply_deltas(ops.linear(rf, deltas_fc.weight, deltas_fc.bias), boxes, w6, rng, clamp)
        # FFN in line: same k-ordered fma chains -> the fused kernel reproduces the per-op chain to the last bits
        got_obj, got_logits, got_pred = ops.stage_tail(obj, (lin1, lin2), norm3, cls_layers, reg_layers, logits_fc, deltas_fc,
                                                        boxes, w6, rng, clamp, split_ffn=False)
        torch.testing.assert_close(got_obj, o2, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(got_logits, lg, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(got_pred, pr, rtol=1e-5, atol=1e-5)
        # FFN split over its hidden slices (the default): F / 128 partial chains added in slice order -- equal to rounding,
        # and the same bits on every run
        got_obj, got_logits, got_pred = ops.stage_tail(obj, (lin1, lin2), norm3, cls_layers, reg_layers, logits_fc, deltas_fc,
                                                        boxes, w6, rng, clamp)
        torch.testing.assert_close(got_obj, o2, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(got_logits, lg, rtol=5e-5, atol=5e-5)
        torch.testing.assert_close(got_pred, pr, rtol=5e-5, atol=5e-5)
        again = ops.stage_tail(obj, (lin1, lin2), norm3, cls_layers, reg_layers, logits_fc, deltas_fc, boxes, w6, rng, clamp)
        assert torch.equal(again[0], got_obj) and torch.equal(again[1], got_logits) and torch.equal(again[2], got_pred)
        # and torch float64
        d = lambda m: (m.weight.double(), None if m.bias is None else m.bias.double())
        x = obj.double()
        h = F.relu(F.linear(x, *d(lin1)))
        x = F.layer_norm(x + F.linear(h, *d(lin2)), (C,), norm3.weight.double(), norm3.bias.double(), norm3.eps)
        torch.testing.assert_close(got_obj.double(), x, rtol=5e-5, atol=5e-5)
        c = x
        for m, n in cls_layers:
            c = F.relu(F.layer_norm(F.linear(c, m.weight.double()), (C,), n.weight.double(), n.bias.double(), n.eps))
        torch.testing.assert_close(got_logits.double(), F.linear(c, *d(logits_fc)), rtol=1e-4, atol=1e-4)
