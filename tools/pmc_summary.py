"""Average the counters of one kernel over its dispatches from rocprofv3 --pmc csv output.
python tools/pmc_summary.py <dir> <kernel-substring>"""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[1:] if len(v) > 2 else v  # drop the first (cold) dispatch
    print(f"{k:32s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    ts = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    if ts:
        ts = ts[1:] if len(ts) > 2 else ts
        print(f"{'duration_ns':32s} n={len(ts):3d} mean={sum(ts) / len(ts):.6g}")
