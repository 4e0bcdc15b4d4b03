#!/usr/bin/env python3
"""Time the LC image branch (VoVNet-99 + FPN + img_convs) under a few execution modes (developer tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import synthetic, workloads  # noqa: E402


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    torch.manual_seed(0)
    m = workloads.build("srfdet_voxel_nusc_LC", 200).eval().cuda()
    img = torch.from_numpy(synthetic.camera_images(3000)).cuda()[0]
    bb, neck = m.img_backbone, m.img_neck

    def run(x):
        with torch.no_grad():
            return neck(list(bb(x).values()))

    print("fp32 nchw            %.1f ms" % timeit(lambda: run(img)))
    torch.backends.cudnn.benchmark = True
    print("fp32 nchw benchmark  %.1f ms" % timeit(lambda: run(img)))
    bb.to(memory_format=torch.channels_last)
    neck.to(memory_format=torch.channels_last)
    xcl = img.contiguous(memory_format=torch.channels_last)
    print("fp32 nhwc benchmark  %.1f ms" % timeit(lambda: run(xcl)))
    for dt in (torch.bfloat16, torch.float16):
        def f():
            with torch.autocast("cuda", dtype=dt):
                return run(xcl)
        print(f"{dt} nhwc autocast %.1f ms" % timeit(f))
        ref = run(xcl)
        out = f()
        print("   max rel err vs fp32:", max(((a.float() - b).abs().max() / b.abs().max()).item() for a, b in zip(out, ref)))


if __name__ == "__main__":
    main()
