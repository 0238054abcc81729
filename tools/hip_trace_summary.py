#!/usr/bin/env python3
"""tools/hip_trace_summary.py DIR [MARKER_KERNEL_SUBSTRING] -- host side of a rocprofv3 --kernel-trace --hip-trace run:
per HIP API function call count and mean / median duration (us), and, for the executions of a latency shape, the host
time between the hipEventRecord before an exec and the one after it (the launch calls of one fwa_plan_exec)."""
import collections
import csv
import glob
import os
import statistics
import sys

api = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
api.sort()
by = collections.defaultdict(list)
for s, e, fn in api:
    by[fn].append((e - s) / 1e3)
print("HIP API calls (host durations, us)")
for fn, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {fn:36s} calls {len(v):7d}  mean {sum(v) / len(v):8.2f}  median {statistics.median(v):8.2f}  total {sum(v) / 1e3:9.3f} ms")
# windows: hipEventRecord ... launches ... hipEventRecord with only launch-type calls in between
launchy = ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtLaunchKernel", "hipExtModuleLaunchKernel", "hipLaunchKernelGGL")
wins = []
i = 0
while i < len(api):
    if api[i][2] == "hipEventRecord":
        j = i + 1
        n_launch = 0
        others = collections.Counter()
        while j < len(api) and api[j][2] != "hipEventRecord":
            if api[j][2] in launchy:
                n_launch += 1
            else:
                others[api[j][2]] += 1
            j += 1
        if j < len(api) and n_launch:
            wins.append(((api[j][0] - api[i][1]) / 1e3, n_launch, tuple(sorted(others.items()))))
        i = j
    else:
        i += 1
groups = collections.defaultdict(list)
for us, nl, oth in wins:
    groups[(nl, oth)].append(us)
print("host time between the event records around an exec (us), by launches per exec and the other HIP calls inside")
for (nl, oth), v in sorted(groups.items(), key=lambda kv: -len(kv[1]))[:12]:
    print(f"  launches {nl:3d}  execs {len(v):5d}  median {statistics.median(v):8.2f}  min {min(v):8.2f}  other calls {dict(oth)}")
