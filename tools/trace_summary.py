#!/usr/bin/env python3
"""tools/trace_summary.py DIR -- per-kernel call counts and mean / min / max duration (us) from rocprofv3 kernel_trace CSVs."""
import csv, glob, os, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:72s} calls {len(v):6d}  mean {sum(v)/len(v):9.2f} us  min {min(v):9.2f}  max {max(v):9.2f}  total {sum(v)/1e3:9.3f} ms")
