#!/usr/bin/env python3
"""Micro-benchmark of srf_conv1x1 on the OSA concat shapes of VoVNet-99 at 6 x 928 x 1600 (developer tool):
fused (no concat, BN + ReLU epilogue) against torch.cat -> conv2d (rocBLAS) -> BN -> ReLU."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import dense  # noqa: E402

SHAPES = [("stage2", 6, (128,) + (128,) * 5, 256, 232, 400), ("stage3", 6, (512,) + (160,) * 5, 512, 116, 200),
          ("stage4", 6, (768,) + (192,) * 5, 768, 58, 100), ("fpn_lat", 6, (768,), 256, 58, 100)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, N, chans, Cout, H, W in SHAPES:
        xs = [torch.randn(N, c, H, W, generator=g).to(dev) for c in chans]
        K = sum(chans)
        conv = nn.Conv2d(K, Cout, 1, bias=False).to(dev)
        bn = nn.BatchNorm2d(Cout).to(dev).eval()
        flops = 2.0 * N * H * W * K * Cout
        with torch.no_grad():
            t_f = timeit(lambda: dense.conv1x1_cat_bn_act(conv, bn, True, xs))
            t_t = timeit(lambda: torch.relu_(bn(conv(torch.cat(xs, 1) if len(xs) > 1 else xs[0]))))
            t_g = timeit(lambda: conv(xs[0] if len(xs) == 1 else cat)) if (cat := torch.cat(xs, 1)) is not None else 0
        print(f"{name}: K={K} Cout={Cout} HW={H * W}: fused {t_f:8.1f} us ({flops / t_f / 1e6:6.1f} TF/s)   torch cat+conv+bn+relu {t_t:8.1f} us"
              f"   rocBLAS conv alone {t_g:8.1f} us ({flops / t_g / 1e6:6.1f} TF/s)")


if __name__ == "__main__":
    main()
