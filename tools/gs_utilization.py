"""Compaction efficiency of the compacted-offset sparse conv: useful pairs / issued MFMA row slots, per level of one synthetic
nuScenes frame, for tile heights TM and group sizes G.  python tools/gs_utilization.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srfdet3d_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
pts = torch.from_numpy(synthetic.nuscenes_sweep(2000, 30000)).to(dev)
_, c, _, _ = ops.hard_voxelize(pts, [0.075, 0.075, 0.2], list(synthetic.NUSC_RANGE), 10, 160000)
idx = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32, device=dev), c], 1).contiguous()
shape = [41, 1472, 1472]
idx = idx[ops.spatial_order(idx, shape, 1)].contiguous()
specs = [(16, [1, 1, 1]), (32, [1, 1, 1]), (64, [0, 1, 1]), (128, None)]
for lvl, (C, pad) in enumerate(specs, 1):
    table = ops.coord_table_build(idx, shape, 1)
    nbr, cnt = ops.rulebook_subm(idx, shape, [3, 3, 3], table)
    A = idx.shape[0]
    has = (nbr[:, :A] >= 0)
    P = int(has.sum())
    line = [f"L{lvl} C={C} A={A} pairs/row={P / A:.2f}"]
    for TM in (32, 64, 88, 120, 152, 176):
        nt = (A + TM - 1) // TM
        padded = torch.zeros(27, nt * TM, dtype=torch.bool, device=dev)
        padded[:, :A] = has
        n = padded.view(27, nt, TM).sum(-1)
        for G in (16, 32, 64):
            slots = ((n + G - 1) // G * G).sum().item()
            line.append(f"TM{TM}/G{G}:{P / slots:.2f}")
    print("  ".join(line), flush=True)
    if pad is None:
        break
    idx, _, _, _, shape = ops.rulebook_strided(idx, shape, 1, [3, 3, 3], [2, 2, 2], pad)
