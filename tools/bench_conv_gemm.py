import os, sys, torch
sys.path.insert(0, "/root/repo")
from srfdet3d_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x0 = torch.randn(64 << 20, device=dev)
for _ in range(20): x0.mul_(1.0)
for (N, H, W, Cin, Cout, s) in [(6, 464, 800, 64, 128, 2), (6, 464, 800, 64, 128, 1), (1, 184, 184, 128, 256, 2), (6, 232, 400, 128, 128, 2)]:
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    pg = ops.pack_conv_gemm_weights(w)
    y = ops.conv_gemm_nhwc(x, pg, Cout, (3, 3), s, 1)
    t = timeit(lambda: ops.conv_gemm_nhwc(x, pg, Cout, (3, 3), s, 1, out=y))
    fl = 2.0 * y.numel() * Cin * 9
    # the same GEMM without the im2col: M = output pixels, K = 9 Cin
    M = y.shape[0] * y.shape[1] * y.shape[2]
    xa = torch.randn(1, 1, M, 9 * Cin, generator=g).to(dev)
    w1 = (torch.randn(Cout, 9 * Cin, generator=g) / (3 * Cin ** 0.5)).to(dev)
    p1 = ops.pack_conv1x1_nhwc_weights(w1)
    y1 = ops.conv1x1_nhwc(xa, p1, Cout)
    t1 = timeit(lambda: ops.conv1x1_nhwc(xa, p1, Cout, out=y1))
    print(f"{Cin}->{Cout} s{s} @{N}x{H}x{W}: conv_gemm {t:8.1f} us {fl / t / 1e6:6.1f} TF   plain GEMM same M,K {t1:8.1f} us {fl / t1 / 1e6:6.1f} TF", flush=True)
