#!/usr/bin/env python3
"""Ordered kernel list of ONE graph-replayed frame out of a rocprofv3 --kernel-trace CSV of bench.py.

    python tools/frame_kernel_list.py <dir with *_kernel_trace.csv> [anchor kernel substring = srf_hv_insert_k] > frame.txt

Prints `start offset (us)  duration (us)  kernel name` from 250 us before the anchor launch in the middle of the trace to the next one (a
timed replay), then the number of launches and of those whose name does not start with `srf_` (torch glue)."""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "srf_hv_insert_k"
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel trace under " + root)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(marks) < 3:
        sys.exit("fewer than three frames in the trace")
    mid = len(marks) // 2   # the middle of the run is inside the timed replays (the last frames are bench.py's eager measurement passes)
    a, b = marks[mid], marks[mid + 1]
    t0 = rows[a][0]
    first = a
    while first > 0 and rows[first - 1][0] > t0 - 250000:
        first -= 1
    n = glue = 0
    for s, e, name in rows[first:b]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:6.1f} {name[:150]}")
        if s >= t0:
            n += 1
            glue += 0 if name.lstrip("void ").startswith("srf_") else 1
    print(f"# launches from the anchor on: {n}; not srf_*: {glue}; frame span {(rows[b][0] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
