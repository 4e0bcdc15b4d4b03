"""srf_conv1x1_nhwc over (M, K, Cout) to see where its rate drops (developer tool).  python tools/bench_gemm_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x0 = torch.randn(64 << 20, device=dev)
for _ in range(20):
    x0.mul_(1.0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in (556800, 139200):
    for K in (128, 256, 576, 768, 1728):
        for Cout in (64, 128, 256, 512):
            if M * K * 4 > 3.5e9:
                continue
            x = torch.randn(1, 1, M, K, generator=g).to(dev)
            w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(dev)
            p = ops.pack_conv1x1_nhwc_weights(w)
            y = ops.conv1x1_nhwc(x, p, Cout)
            t = timeit(lambda: ops.conv1x1_nhwc(x, p, Cout, out=y))
            fl = 2.0 * M * K * Cout
            by = 4.0 * M * (K + Cout)
            print(f"M={M:7d} K={K:5d} Cout={Cout:4d}: {t:8.1f} us  {fl / t / 1e6:6.1f} TF  {by / t / 1e3:7.1f} GB/s algorithmic", flush=True)
            del x, y
