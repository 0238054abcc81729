#!/usr/bin/env python3
"""tools/kernel_evidence_table.py RAW [RAW ...] -- the table of profiles/round6/kernel_evidence_all_families.txt from the raw
summaries tools/run_kernel_evidence.sh appends (one block per size: the one_exec JSON line, tools/trace_summary.py rows,
tools/pmc_summary.py rows for FETCH_SIZE and WRITE_SIZE).

Per kernel: launches per exec, mean duration, FETCH_SIZE and WRITE_SIZE per launch (KB, as rocprofv3 reports them) and the
L2<->fabric traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE counts half the bytes of a wide read
stream, MI355X_MICROARCH.md; the rule is re-checked in every block on the one-launch kernels, which read each byte once).
Per size: traffic per exec / algorithmic bytes (16 B x 2^32 samples), the HIP-event time of an exec (median of the three of
the --stats pass) and frac = 16 B x 2^32 / that time / 8 TB/s."""
import json
import re
import sys

ALGO = 16.0 * (1 << 32)


def blocks(paths):
    cur = None
    for p in paths:
        for line in open(p):
            line = line.rstrip("\n")
            if line.startswith("== "):
                cur = {"title": line[3:], "kernels": {}, "exec": None}
                yield cur
            elif cur is None or line.startswith("done"):
                continue
            elif line.startswith("{"):
                cur["exec"] = json.loads(line)
            else:
                m = re.match(r"(?:void )?(fwa::.+?)\s+calls\s+(\d+)\s+mean\s+([0-9.]+) us", line)
                if m:
                    cur["kernels"].setdefault(m.group(1)[:60], {}).update(calls=int(m.group(2)), mean_us=float(m.group(3)))
                    continue
                m = re.match(r"(?:void )?(fwa::.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+)\s+sum\s+\S+\s+per dispatch\s+(\S+)", line)
                if m:
                    cur["kernels"].setdefault(m.group(1)[:60], {})[m.group(2)] = float(m.group(4))


def main():
    rows = list(blocks(sys.argv[1:]))
    print("# one rocprofv3 evidence pass per kernel family (tools/run_kernel_evidence.sh; 2^32 samples = 32 GiB per exec, 3 execs)")
    print("# size | kernel | launches/exec | mean us | FETCH KB/launch | WRITE KB/launch | (2F+W) KB/launch")
    for b in rows:
        ex = b["exec"] or {}
        ms = sorted(ex.get("exec_ms", [0]))[len(ex.get("exec_ms", [0])) // 2]
        traffic = 0.0
        print(f"\n## {b['title'].split(':')[0]}   path {ex.get('path')} factors {ex.get('factors')} launches/exec {ex.get('launches_per_exec')} chains {ex.get('streams')}")
        for k, v in sorted(b["kernels"].items(), key=lambda kv: -kv[1].get("mean_us", 0) * kv[1].get("calls", 0)):
            if "calls" not in v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
                continue
            per_exec = v["calls"] / 3
            t = 2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
            traffic += t * 1024 * per_exec
            print(f"   {k:52s} {per_exec:8.0f} {v['mean_us']:10.2f} {v['FETCH_SIZE']:14.1f} {v['WRITE_SIZE']:14.1f} {t:14.1f}")
        sum_us = sum(v.get("mean_us", 0) * v.get("calls", 0) / 3 for v in b["kernels"].values())
        chains = max(1, int(ex.get("streams") or 1))
        print(f"   => traffic per exec {traffic / 1e9:8.2f} GB = {traffic / ALGO:5.3f} x algorithmic;  exec (HIP events, median of 3, under the "
              f"kernel trace) {ms:8.3f} ms = frac {ALGO / (ms * 1e-3) / 8e12 if ms else 0:5.3f};  sum of kernel durations / chains "
              f"{sum_us / chains / 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
