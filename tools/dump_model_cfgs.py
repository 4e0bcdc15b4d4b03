#!/usr/bin/env python3
"""Evaluate the unchanged reference config files and store their `model` dicts (values only) as JSON under
srfdet3d_amd/workloads/, so that bench.py and the GPU tests -- which run where /root/reference does not exist --
build exactly the models the reference configs describe.  tests/test_configs.py re-evaluates the configs whenever
the reference is present and fails if a JSON has drifted.

usage: python tools/dump_model_cfgs.py [--ref /root/reference]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srfdet3d_amd.compat.config import Config  # noqa: E402

CONFIGS = {
    "srfdet_voxel_nusc_L": "configs/nus/srfdet_voxel_nusc_L.py",
    "srfdet_voxel_nusc_LC": "configs/nus/srfdet_voxel_nusc_LC.py",
    "srfdet_voxel_kitti_L": "configs/kitti/srfdet_voxel_kitti_L.py",
    "srfdet_dvoxel_waymo_L": "configs/waymo/srfdet_dvoxel_waymo_L.py",
    "srfdet_pillar_nusc_L": "configs/nus/srfdet_pillar_nusc_L.py",
    "srfdet_voxel_kitti_LC": "configs/kitti/srfdet_voxel_kitti_LC.py",
    "srfdet_pillar_v299_nusc_LC": "configs/nus/srfdet_pillar_v299_nusc_LC.py",
    "srfdet_pillar_r50_nusc_LC": "configs/nus/srfdet_pillar_r50_nusc_LC.py",
    "srfdet_voxel_r50_nusc_LC": "configs/nus/srfdet_voxel_r50_nusc_LC.py",
    "srfdet_dvoxel_nusc_L": "configs/others/srfdet_dvoxel_nusc_L.py",
    "srfdet_dvoxel_waymo_LC": "configs/others/srfdet_dvoxel_waymo_LC.py",
}


def plain(o):
    if isinstance(o, dict):
        return {k: plain(v) for k, v in o.items()}
    if isinstance(o, tuple):
        return {"__tuple__": [plain(v) for v in o]}
    if isinstance(o, list):
        return [plain(v) for v in o]
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "srfdet3d_amd", "workloads")
    for name, rel in CONFIGS.items():
        cfg = Config.fromfile(os.path.join(a.ref, rel))
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(dict(source=rel, model=plain(cfg.model)), f, indent=1, sort_keys=True)
        print("wrote", name)


if __name__ == "__main__":
    main()
