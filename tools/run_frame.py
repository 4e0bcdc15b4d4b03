#!/usr/bin/env python3
"""Run a few synthetic frames through a workload on cuda:0 and print stage timings (developer tool)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srfdet3d_amd as S  # noqa: E402
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402


def randomize_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="srfdet_voxel_nusc_L")
    ap.add_argument("--np", type=int, default=200)
    ap.add_argument("--frames", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    model = workloads.build(a.workload, a.np).eval()
    randomize_bn(model)
    model = model.cuda()
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    for i in range(a.frames):
        pts = torch.from_numpy(synthetic.nuscenes_sweep(2000 + i)).cuda()
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            feats = model.extract_point_features([pts])
            torch.cuda.synchronize()
            t1 = time.time()
            logits, boxes = model.bbox_head(None, feats, metas)
            torch.cuda.synchronize()
            t2 = time.time()
            res = model.bbox_head.get_bboxes(logits, boxes, metas)
            torch.cuda.synchronize()
            t3 = time.time()
        print(f"frame {i}: feats {1e3*(t1-t0):.2f} ms, head {1e3*(t2-t1):.2f} ms, decode+nms {1e3*(t3-t2):.2f} ms, "
              f"dets {len(res[0][1])}, feats {[tuple(f.shape) for f in feats]}")


if __name__ == "__main__":
    main()
