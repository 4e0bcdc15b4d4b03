#!/usr/bin/env python3
"""Time SECONDCustom + FPN on the nuScenes BEV map under MIOpen execution modes (developer tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import workloads  # noqa: E402


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    torch.manual_seed(0)
    m = workloads.build("srfdet_voxel_nusc_L", 200).eval().cuda()
    x = torch.randn(1, 256, 184, 184, device="cuda")
    x = x * (torch.rand_like(x) < 0.3)

    def run(inp):
        with torch.no_grad():
            return m.pts_neck(m.pts_backbone(inp))

    ref = run(x)
    print("nchw                 %.3f ms" % timeit(lambda: run(x)))
    torch.backends.cudnn.benchmark = True
    print("nchw + benchmark     %.3f ms" % timeit(lambda: run(x)))
    torch.backends.cudnn.benchmark = False
    m.pts_backbone.to(memory_format=torch.channels_last)
    m.pts_neck.to(memory_format=torch.channels_last)
    xc = x.contiguous(memory_format=torch.channels_last)
    print("nhwc                 %.3f ms" % timeit(lambda: run(xc)))
    torch.backends.cudnn.benchmark = True
    print("nhwc + benchmark     %.3f ms" % timeit(lambda: run(xc)))
    out = run(xc)
    print("max rel diff vs nchw:", max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(out, ref)))


if __name__ == "__main__":
    main()
