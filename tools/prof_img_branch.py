"""A few eager passes of the LC camera branch (VoVNet-99 -> FPN -> the head's img_convs) and nothing else: the target of
the `rocprofv3 --pmc` passes of tools/measure_traffic.py (srf_wino3x3_k, srf_conv1x1_nhwc_k: counters averaged over every
launch of a frame, like bench.py's aggregated `roofline`).
python tools/prof_img_branch.py [passes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, randomize_bn  # noqa: E402
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(0)
model = workloads.build(WORKLOADS["nusc_LC"]["cfg"], 200).eval()
randomize_bn(model)
model = model.cuda()
img = torch.from_numpy(synthetic.camera_images(3000)).cuda()
metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in synthetic.camera_rig()])]
with torch.no_grad():
    for _ in range(passes):
        feats = model.extract_img_feat(img, metas)
        model.bbox_head._img_convs_only(feats)
torch.cuda.synchronize()
print("done", [tuple(f.shape) for f in feats])
