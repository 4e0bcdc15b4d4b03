#!/usr/bin/env python3
"""ms per frame of bench.py's roofline kernel families INSIDE the timed hipGraphs, from a rocprofv3 --kernel-trace --stats
summary of the bench command (the camera graph shares the chip with the BEV half there, so these are larger than the serial
HIP-event times of `roofline`).

    python tools/in_graph_summary.py profiles/r03_bench_nuscLC_np200_kernel_stats.csv nusc_LC 88 [out.json]

`launches per frame` of the reference family (srf_wino43_mm_k: 88 on LC) turns call counts into frames: every graph replay,
validation replay and warm-up pass of the traced process is a frame like any other.  bench.py's eager post-timing passes of the camera
branch (3, for the per-launch HIP events of `roofline`) run the same launches serially, not beside the BEV half: they are a few of ~80
frames in the trace and bias `in_graph_ms_per_frame` low by that share; `_frames_in_trace` says how many frames the sums cover and
`_eager_passes_in_trace` how many of them were those serial passes."""
import csv
import json
import os
import sys

FAMILIES = {   # bench.py name -> kernel-name substrings
    "wino43mm": ("srf_wino43_mm_k<",),
    "wino43xf": ("srf_wino43_xform_k",),
    "wino3x3": ("srf_wino3x3_k<", "srf_wino3x3_mixed_k<"),
    "conv1x1": ("srf_conv1x1_nhwc_k<1, 1, 4, false>", "srf_conv1x1_nhwc_k<2, 2, 3, false>", "srf_conv1x1_nhwc_k<4, 4, 1, false>",
                "srf_conv1x1_nhwc_mixed_k", "srf_gemm_direct_k<"),
    "gemmsplit": ("srf_gemm_split_k<",),
}


def source_id():
    """sha of the kernel sources (bench.py quotes this file only for the build it was measured on, as for the traffic files)"""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "srfdet3d_amd", "csrc", "*.h*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    path, workload, per_frame = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(path), os.path.basename(path).split("_")[0] + "_in_graph_summary.json")
    rows = list(csv.DictReader(open(path)))
    tot = {k: [0, 0.0] for k in FAMILIES}
    for r in rows:
        for k, subs in FAMILIES.items():
            if any(s in r["Name"] for s in subs):
                tot[k][0] += int(r["Calls"])
                tot[k][1] += float(r["TotalDurationNs"])
    frames = tot["wino43mm"][0] / per_frame if tot["wino43mm"][0] else 0
    res = {}
    if os.path.exists(out):
        res = json.load(open(out))
    res[workload] = {k: round(v[1] / 1e6 / frames, 3) for k, v in tot.items() if frames and v[0]}
    res[workload]["_frames_in_trace"] = round(frames, 2)
    res[workload]["_eager_passes_in_trace"] = 3
    res[workload]["_source"] = os.path.basename(path)
    res["kernel_source_sha16"] = source_id()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res[workload]))


if __name__ == "__main__":
    main()
