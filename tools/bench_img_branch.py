#!/usr/bin/env python3
"""Time the LC image branch (VoVNet-99 + FPN) in fp32 under a few MIOpen settings (developer tool).
usage: python tools/bench_img_branch.py [--benchmark] [--nhwc]   (MIOPEN_* environment variables pass through)"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--benchmark", action="store_true")
    ap.add_argument("--nhwc", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    m = workloads.build("srfdet_voxel_nusc_LC", 200).eval().cuda()
    img = torch.from_numpy(synthetic.camera_images(3000)).cuda()[0]
    bb, neck = m.img_backbone, m.img_neck
    torch.backends.cudnn.benchmark = a.benchmark
    if a.nhwc:
        bb.to(memory_format=torch.channels_last)
        neck.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)

    def run():
        with torch.no_grad():
            return neck(list(bb(img).values()))

    print(f"benchmark={a.benchmark} nhwc={a.nhwc} env={ {k: v for k, v in os.environ.items() if k.startswith('MIOPEN')} }: "
          f"{timeit(run):.1f} ms")


if __name__ == "__main__":
    main()
