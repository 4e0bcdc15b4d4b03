#!/usr/bin/env python3
"""Micro-benchmark of the sparse-conv kernel on the real active sets of one synthetic frame (developer tool).
usage: python tools/bench_spconv.py [--reps 20] [--levels 4]   (prints per-level time, algorithmic and executed TFLOP/s)"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--levels", default="1,2,3,4")
    ap.add_argument("--points", type=int, default=30000)
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--unpacked", action="store_true")
    ap.add_argument("--no-tiles", action="store_true", help="equal-height tiles instead of work-balanced row ranges")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pts = torch.from_numpy(synthetic.nuscenes_sweep(2000, a.points)).to(dev)
    _, c, _, _ = ops.hard_voxelize(pts, [0.075, 0.075, 0.2], list(synthetic.NUSC_RANGE), 10, 160000)
    idx = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32, device=dev), c], 1).contiguous()
    shape = [41, 1472, 1472]
    if not a.no_sort:
        idx = idx[ops.spatial_order(idx, shape, 1)].contiguous()
    specs = [(16, [1, 1, 1]), (32, [1, 1, 1]), (64, [0, 1, 1]), (128, None)]
    want = [int(x) for x in a.levels.split(",")]
    g = torch.Generator(device="cpu").manual_seed(0)
    for lvl, (C, pad) in enumerate(specs, 1):
        table = ops.coord_table_build(idx, shape, 1)
        nbr, cnt = ops.rulebook_subm(idx, shape, [3, 3, 3], table)
        A = idx.shape[0]
        if lvl in want:
            f = torch.randn(A, C, generator=g).to(dev)
            W = (torch.randn(27, C, C, generator=g) * 0.05).to(dev)
            al, be = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev)
            pk = ops.pack_spconv_weights(W) if (C >= 32 and not a.unpacked) else None
            tiles = ops.spconv_tiles(nbr) if (pk is not None and ops.spconv_tiles_wanted(C, C) and not a.no_tiles) else None
            for _ in range(3):
                ops.spconv_fwd(f, W, nbr, al, be, f, True, packed=pk, tiles=tiles)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.reps):
                ops.spconv_fwd(f, W, nbr, al, be, f, True, packed=pk, tiles=tiles)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            P = int(cnt.sum().item())
            print(f"L{lvl}: A={A} C={C} pairs={P} ({P / A:.1f}/row)  {us:8.1f} us  algorithmic {2 * P * C * C / us / 1e6:7.2f} TF/s  "
                  f"executed(27 taps) {2 * 27 * A * C * C / us / 1e6:7.2f} TF/s")
        if pad is None:
            break
        idx, _, _, _, shape = ops.rulebook_strided(idx, shape, 1, [3, 3, 3], [2, 2, 2], pad)


if __name__ == "__main__":
    main()
