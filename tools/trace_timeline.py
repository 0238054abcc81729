#!/usr/bin/env python3
"""tools/trace_timeline.py DIR [N] -- the last N kernel records of a rocprofv3 kernel trace in start order: start offset,
duration and the gap to the previous kernel's end (us)."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[-n:]
t0, prev = rows[0][0], None
for s, e, k in rows:
    gap = "" if prev is None else f"gap {(s - prev) / 1e3:7.2f}"
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f}  {gap:14s} {k}")
    prev = e
