"""Winograd F(4x4, 3x3) in f32, emulated on the CPU with exactly the arithmetic of csrc/wino43.hip, against a float64 direct
convolution -- the accuracy table behind the decision to run the stride-1 3x3 layers (vovnet.py:116-133, :180-216;
srfdet_head.py:404-416; second_custom.py:41-63) on `srf_wino43`.

Emulated: U = G g G^T in double, rounded once; V = B^T d B as the kernel's fma sequence (two 1-D passes, f32); the products
accumulated over the input channels as one ascending f32 chain per (frequency, tile, output channel); Y = A^T M A as the
kernel's sequence.  Inputs are post-ReLU Gaussians, weights He-scaled: the statistics the layers see.

Result (printed with -s): max error / map max = 4e-6 (Cin = 64) ... 2.2e-5 (Cin = 1024), F(2x2, 3x3) on the same data
4-6e-7.  Bar per layer: 3e-5 of the map's maximum (tests/test_gpu_conv.py holds the HIP kernel to the same bar); the camera
branch end to end stays inside 2e-4 (tests/test_gpu_conv.py, tests/test_gpu_fixtures_r3.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

f32 = np.float32
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
             dtype=np.float64)


def _fma(a, b, c):
    """f32 fused multiply-add: one rounding (computed in float64, exact for f32 operands, rounded once)."""
    return (np.float64(a) * np.float64(b) + np.float64(c)).astype(f32)


def bt_1d(x):
    """B^T x along axis 0 (6 -> 6), the W43_BT sequence."""
    x0, x1, x2, x3, x4, x5 = x
    a = _fma(f32(-4), x2, x4)
    b = _fma(f32(-4), x1, x3)
    c = (x4 - x2).astype(f32)
    e = (x3 - x1).astype(f32)
    v0 = _fma(f32(4), x0, _fma(f32(-5), x2, x4))
    v5 = _fma(f32(4), x1, _fma(f32(-5), x3, x5))
    return np.stack([v0, (a + b).astype(f32), (a - b).astype(f32), _fma(f32(2), e, c), _fma(f32(-2), e, c), v5])


def at_1d(m):
    """A^T m along axis 0 (6 -> 4), the W43_AT sequence."""
    m0, m1, m2, m3, m4, m5 = m
    s1, d1 = (m1 + m2).astype(f32), (m1 - m2).astype(f32)
    s2, d2 = (m3 + m4).astype(f32), (m3 - m4).astype(f32)
    z0 = ((m0 + s1).astype(f32) + s2).astype(f32)
    return np.stack([z0, _fma(f32(2), d2, d1), _fma(f32(4), s2, s1), (_fma(f32(8), d2, d1) + m5).astype(f32)])


def wino43_emulated(x, w):
    """x (C, H, W) f32 with H, W multiples of 4; w (K, C, 3, 3) f32 -> (K, H, W) f32."""
    C, H, W = x.shape
    K = w.shape[0]
    ty, tx = H // 4, W // 4
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    U = np.einsum("ia,kcab,jb->ijkc", G, w.astype(np.float64), G).astype(f32)          # (6, 6, K, C)
    d = np.empty((6, 6, C, ty, tx), f32)
    for py in range(6):
        for px in range(6):
            d[py, px] = xp[:, py:py + 4 * ty:4, px:px + 4 * tx:4]
    V = bt_1d(d)                                   # vertical pass (over py)
    V = bt_1d(V.transpose(1, 0, 2, 3, 4)).transpose(1, 0, 2, 3, 4)   # horizontal pass (over px)
    M = np.zeros((6, 6, K, ty, tx), f32)
    for c in range(C):                             # one ascending fma chain per accumulator (the MFMA's k order)
        M = _fma(U[:, :, :, c, None, None], V[:, :, None, c], M)
    Z = at_1d(M)                                   # column stage: over the frequency rows
    Y = at_1d(Z.transpose(1, 0, 2, 3, 4)).transpose(1, 0, 2, 3, 4)   # row stage
    out = np.empty((K, H, W), f32)
    for i in range(4):
        for j in range(4):
            out[:, i::4, j::4] = Y[i, j]
    return out


SHAPES = [  # (Cin, Cout, what)
    (64, 64, "VoVNet stem_2"), (128, 128, "stage 2 / SECOND block 1"), (160, 160, "stage 3"), (256, 160, "stage 3 first layer"),
    (192, 192, "stage 4"), (768, 192, "stage 4 first layer"), (224, 224, "stage 5"), (1024, 224, "stage 5 first layer"),
    (256, 256, "image FPN / SECOND block 2"), (256, 128, "img_convs"),
]


@pytest.mark.parametrize("Cin,Cout,what", SHAPES)
def test_f43_f32_error_per_layer_shape(Cin, Cout, what):
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(Cin, 16, 16, generator=g, dtype=torch.float64).relu()
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (9 * Cin)) ** 0.5
    ref = F.conv2d(x[None], w, padding=1)[0].numpy()
    out = wino43_emulated(x.numpy().astype(f32), w.numpy().astype(f32))
    # the f32-rounded inputs are what both see
    ref32 = F.conv2d(x.float().double()[None], w.float().double(), padding=1)[0].numpy()
    err = np.abs(out - ref32).max() / np.abs(ref32).max()
    print(f"F(4x4,3x3) f32  {Cin:5d} -> {Cout:4d}  ({what}): max err / map max = {err:.2e}")
    assert err <= 3e-5, err
    assert np.abs(ref - ref32).max() / np.abs(ref).max() < 1e-6


def test_transforms_are_the_winograd_identity():
    """B^T, G, A^T as written above reproduce the 3-tap correlation exactly in float64 (catches a wrong sign or row order in the
    emulation, which mirrors the kernel's macros line by line)."""
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(6), rng.standard_normal(3)
    Bt = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
    At = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
    y = At @ ((G @ g) * (Bt @ d))
    ref = np.array([d[i:i + 3] @ g for i in range(4)])
    assert np.allclose(y, ref, atol=1e-12)
    # the fma sequences equal the matrices
    x = rng.standard_normal((6, 5)).astype(f32)
    assert np.allclose(bt_1d(list(x)), Bt @ x, atol=1e-5)
    assert np.allclose(at_1d(list(x)), At @ x, atol=1e-5)
