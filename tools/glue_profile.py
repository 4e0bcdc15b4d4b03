"""Where the small torch launches of the static LiDAR frame come from (developer tool): the torch calls of one eager pass of
GraphedFrame._run_bev + _run_head that allocate-and-fill or copy, counted by the srfdet3d_amd source line that issued them
(thin wrappers around the torch entry points record the caller).  python tools/glue_profile.py [workload]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, randomize_bn  # noqa: E402
from srfdet3d_amd import synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402
from srfdet3d_amd.graphs import GraphedFrame  # noqa: E402

COUNT = collections.Counter()
ON = [False]


def _caller():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "srfdet3d_amd" in fr.filename and "glue_profile" not in fr.filename:
            return f"{fr.filename.split('srfdet3d_amd/')[-1]}:{fr.lineno}"
    return "?"


def _wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ON[0]:
            COUNT[(name, _caller())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for n in ("zeros", "full", "cat", "stack", "where", "zeros_like", "full_like", "empty_like", "ones"):
    _wrap(torch, n)
for n in ("copy_", "clone", "contiguous", "fill_", "zero_", "to", "sum", "new_zeros", "new_full", "float", "int", "long", "index_select",
          "__getitem__", "__setitem__"):
    _wrap(torch.Tensor, n)

wl = sys.argv[1] if len(sys.argv) > 1 else "nusc_L"
torch.manual_seed(0)
m = workloads.build(WORKLOADS[wl]["cfg"], 200).eval()
randomize_bn(m)
m = m.cuda()
pts = torch.from_numpy(synthetic.nuscenes_sweep(2000, 30000)).cuda()
metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
gf = GraphedFrame(m)
with torch.no_grad():
    bev, sizes = gf._measure(pts)
    caps = {k: gf._round(v * 1.5) for k, v in sizes.items()}
    static_pts = torch.full((33000, 5), 1e6, device="cuda")
    static_pts[:pts.shape[0]] = pts
    for _ in range(2):
        x, counts = gf._run_bev(static_pts, caps)
        gf._run_head(x, metas, None)
    torch.cuda.synchronize()
    ON[0] = True
    x, counts = gf._run_bev(static_pts, caps)
    gf._run_head(x, metas, None)
    ON[0] = False
for (name, where), n in sorted(COUNT.items(), key=lambda kv: (-kv[1], kv[0])):
    if name in ("__getitem__", "contiguous", "to", "float", "int", "long") and n < 3:
        continue
    print(f"{n:4d}  {name:12s} {where}")
