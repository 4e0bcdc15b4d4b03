#!/usr/bin/env python3
"""srf_conv_wgrad_nhwc against the library routes (MIOpen aten.convolution_backward for 3x3, a rocBLAS TN GEMM for 1x1) on the trainable
layer shapes of config 4 (bs = 2: 12 camera images): microseconds per call and TFLOP/s of the direct weight-gradient FLOPs.
python tools/wgrad_bench.py > profiles/r05_wgrad_bench.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops  # noqa: E402

SHAPES = [  # (what, N, H, W, Cin, Cout, k)
    ("VoVNet stage 4 3x3 (45 layers)", 12, 58, 100, 192, 192, 3),
    ("VoVNet stage 4 concat 1x1", 12, 58, 100, 1728, 768, 1),
    ("VoVNet stage 5 3x3 (15 layers)", 12, 29, 50, 224, 224, 3),
    ("VoVNet stage 5 concat 1x1", 12, 29, 50, 2144, 1024, 1),
    ("image FPN output / img_convs 3x3, level 0", 12, 232, 400, 256, 256, 3),
    ("image FPN output / img_convs 3x3, level 1", 12, 116, 200, 256, 256, 3),
    ("image FPN output / img_convs 3x3, level 2", 12, 58, 100, 256, 256, 3),
    ("image FPN lateral 1x1, level 2", 12, 58, 100, 768, 256, 1),
]


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    g_ = torch.Generator().manual_seed(0)
    print(f"{'layer':46s} {'GFLOP':>8s} {'srf us':>9s} {'TF/s':>7s} {'library us':>11s} {'TF/s':>7s} {'max |diff| / max':>17s}")
    for what, N, H, W, Cin, Cout, k in SHAPES:
        if N * H * W * max(Cin, Cout) * 4 >= (1 << 31):
            N = N // 2
            what += f" (N = {N})"
        x = torch.relu(torch.randn(N, H, W, Cin, generator=g_)).to(dev)
        g = (torch.randn(N, H, W, Cout, generator=g_) * 0.05).to(dev)
        w = torch.zeros(Cout, Cin, k, k, device=dev)
        fl = 2.0 * N * H * W * Cin * Cout * k * k
        reps = 5 if fl > 5e11 else 20
        if k == 3:
            lib = lambda: torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, None, (1, 1), (1, 1), (1, 1), False,
                                                              (0, 0), 1, (False, True, False))[1]
        else:
            lib = lambda: (g.reshape(-1, Cout).t() @ x.reshape(-1, Cin)).view(Cout, Cin, 1, 1)
        ours = lambda: ops.conv_wgrad_nhwc(g, x, k)
        a, b = ours(), lib()
        diff = (a - b).abs().max().item() / b.abs().max().item()
        t_o, t_l = timed(ours, reps), timed(lib, reps)
        print(f"{what:46s} {fl / 1e9:8.1f} {t_o:9.1f} {fl / t_o / 1e6:7.1f} {t_l:11.1f} {fl / t_l / 1e6:7.1f} {diff:17.2e}")


if __name__ == "__main__":
    main()
