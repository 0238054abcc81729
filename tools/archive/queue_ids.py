#!/usr/bin/env python3
"""tools/queue_ids.py DIR -- from a rocprofv3 kernel trace: per consecutive run of FFT pass kernels, which Queue_Ids ran them."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cur, runs = None, []
for r in rows:
    k = r["Kernel_Name"]
    if "k_fill" in k or "copyBuffer" in k:
        if cur: runs.append(cur); cur = None
        continue
    if cur is None: cur = {"q": {}, "t0": int(r["Start_Timestamp"]), "t1": 0}
    cur["q"][r["Queue_Id"]] = cur["q"].get(r["Queue_Id"], 0) + 1
    cur["t1"] = max(cur["t1"], int(r["End_Timestamp"]))
if cur: runs.append(cur)
for i, c in enumerate(runs):
    print(i, "ms %.3f" % ((c["t1"] - c["t0"]) / 1e6), "queues", c["q"])
