"""Kernel, hardware queue and stream of every launch of ONE replayed frame from a rocprofv3 --kernel-trace CSV of bench.py (marks
launches that start before the previous one ended): `python tools/frame_queue_list.py <trace dir>`."""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "srf_hv_insert_k" in r[2]]
mid = len(marks) // 2
a, b = marks[mid], marks[mid + 1]
t0 = rows[a][0]
prev_end = 0
for s, e, name, q, st in rows[a:b]:
    ov = "OVERLAP" if s < prev_end else ""
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:6.1f} q{q} s{st} {name[:60]} {ov}")
    prev_end = max(prev_end, e)
print("span", (rows[b][0] - t0) / 1e3)
