"""Does the camera branch (VoVNet-99 -> FPN -> img_convs on six views) run faster as several independent sub-batches on
several streams than as one batch-6 chain?  A serial chain alternates HBM-bound kernels (Winograd input transform, eSE
affine, pooling) with MFMA-bound ones (Winograd multiply, 1x1 GEMMs); chains of different views out of phase could hide the
first kind under the second.  Developer tool: python tools/split_batch_probe.py [parts ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, randomize_bn  # noqa: E402
from srfdet3d_amd import graphs, synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402

torch.manual_seed(0)
model = workloads.build(WORKLOADS["nusc_LC"]["cfg"], 200).eval()
randomize_bn(model)
model = model.cuda()
img = torch.from_numpy(synthetic.camera_images(3000)).cuda()
metas = [dict(box_type_3d=LiDARInstance3DBoxes, lidar2img=[m for m in synthetic.camera_rig()])]
parts_list = [int(a) for a in sys.argv[1:]] or [1, 2, 3]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for parts in parts_list:
        per = 6 // parts
        chunks = [img[:, i * per:(i + 1) * per].contiguous() for i in range(parts)]
        for overlap in ((False,) if parts == 1 else (False, True)):
            branches = [graphs.GraphedImageBranch(model, overlap=overlap) for _ in range(parts)]
            main = torch.cuda.current_stream()

            def run():
                evs = [b(c, [dict(m) for m in metas])[1] for b, c in zip(branches, chunks)]
                for e in evs:
                    main.wait_event(e)

            ms = timed(run)
            print(f"{parts} sub-batch(es) of {per} views, {'one stream each' if overlap else 'same stream'}: {ms:8.3f} ms", flush=True)
            del branches
            torch.cuda.empty_cache()
