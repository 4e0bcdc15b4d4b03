#!/usr/bin/env python3
"""Where does the host spend the ~14 ms behind the camera-graph launch of an LC frame?  Times every host call of
GraphedImageBranch.__call__ and of the frame around it on steady-state frames, for the camera graph as shipped (coarse FPN levels forked
into parallel graph branches) and with the forks off (SRF_FPN_FORK=0, if the build has that switch).
python tools/lc_launch_probe.py > profiles/r05_lc_launch_probe.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, randomize_bn  # noqa: E402
from srfdet3d_amd import graphs, synthetic, workloads  # noqa: E402
from srfdet3d_amd.compat.boxes import LiDARInstance3DBoxes  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = workloads.build(WORKLOADS["nusc_LC"]["cfg"], 200).eval()
    randomize_bn(model)
    model = model.to(dev)
    model.enable_hip_graphs(img_overlap=True, whole_frame=True)
    frames = [torch.from_numpy(synthetic.nuscenes_sweep(2000 + i, 30000)).to(dev) for i in range(4)]
    metas = [dict(box_type_3d=LiDARInstance3DBoxes)]
    img = torch.from_numpy(synthetic.camera_images(3000)).to(dev)
    metas[0]["lidar2img"] = [m for m in synthetic.camera_rig()]
    with torch.no_grad():
        for i in range(6):
            model.simple_test(img, [frames[i % 4]], metas)
    torch.cuda.synchronize()

    gi = model._graphed_img
    rec = {}

    def timed_call(img_, img_metas):
        key = (tuple(img_.shape), img_.dtype)
        e = gi.entries[key]
        main_s = torch.cuda.current_stream()
        for meta in img_metas:
            meta.update(input_shape=img_.shape[-2:])
        run_on = gi.stream if gi.overlap else main_s
        t = [time.perf_counter()]
        run_on.wait_stream(main_s)
        t.append(time.perf_counter())
        with torch.cuda.stream(run_on):
            e["img"].copy_(img_)
            t.append(time.perf_counter())
            e["graph"].replay()
            t.append(time.perf_counter())
            e["done"].record(run_on)
            t.append(time.perf_counter())
        rec.setdefault("img", []).append([(b - a) * 1e3 for a, b in zip(t, t[1:])])
        return e["feats"], e["done"]

    gi_call = type(gi).__call__
    type(gi).__call__ = lambda self, a, b: timed_call(a, b)
    gf = model._graphed_frame
    gf_call = type(gf).__call__

    def timed_frame(self, *a, **k):
        t0 = time.perf_counter()
        out = gf_call(self, *a, **k)
        rec.setdefault("frame", []).append((time.perf_counter() - t0) * 1e3)
        return out
    type(gf).__call__ = timed_frame
    whole = []
    with torch.no_grad():
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.simple_test(img, [frames[i % 4]], metas)
            torch.cuda.synchronize()
            whole.append((time.perf_counter() - t0) * 1e3)
    type(gi).__call__ = gi_call
    type(gf).__call__ = gf_call
    print(f"frames: {', '.join(f'{w:.2f}' for w in whole)} ms (synchronous)")
    print("camera branch, host ms per call: wait_stream | img.copy_ | graph.replay | event.record")
    for r in rec["img"]:
        print("   " + " | ".join(f"{x:8.3f}" for x in r))
    print("LiDAR frame object (BEV graph + decoder graph + read-back), host ms per call (returns after the read-back = end of frame):")
    print("   " + ", ".join(f"{x:.2f}" for x in rec["frame"]))
    # the graph itself
    g = gi.entries[next(iter(gi.entries))]["graph"]
    try:
        g.enable_debug_mode()
    except Exception:
        pass
    # replay alone, nothing else queued
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        with torch.cuda.stream(gi.stream):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"camera graph alone: replay() returned after {(t1 - t0) * 1e3:.3f} ms, graph done after {(t2 - t0) * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
