#!/usr/bin/env python3
"""tools/make_bench_reference.py DIR ROUND [COMMIT] -- refresh profiles/bench_reference.json from the summaries of ONE profile job
(tools/run_round_profiles.sh writes them to DIR just before it takes the bench line, so that the line, the PMC summaries and
the kernel statistics of a round are one consistent set: VERDICT round 4, item 5).  The file is stamped with the commit the job
ran on (COMMIT, or $FWA_COMMIT: the GPU box has no .git) and with sha256(csrc/tile_1m.h + csrc/kernels_1m.hip), the sources of
the two kernels the traffic figure belongs to: bench.py marks the figure `"traffic_stale": true` when that hash has moved on.

Reads DIR/a_pmc_fetch_summary.txt, a_pmc_write_summary.txt (tools/pmc_summary.py) and a_trace_streams1_summary.txt
(tools/trace_summary.py); keeps the probe-derived entries (measured_floors) of the existing file.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("fft_wgpu_amd/csrc/tile_1m.h", "fft_wgpu_amd/csrc/kernels_1m.hip")   # bench.py hashes the same list


def kernel_source_sha256(root=ROOT):
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(root, rel), "rb").read())
    return h.hexdigest()


def per_dispatch(path, kernel):
    for line in open(path):
        if kernel in line:
            return float(line.split("per dispatch")[1]), int(re.search(r"dispatches\s+(\d+)", line).group(1))
    raise SystemExit(f"{kernel} not in {path}")


def mean_us(path, kernel):
    for line in open(path):
        if kernel in line:
            return float(re.search(r"mean\s+([0-9.]+) us", line).group(1)), int(re.search(r"calls\s+(\d+)", line).group(1))
    raise SystemExit(f"{kernel} not in {path}")


def main():
    d, rnd = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("FWA_COMMIT", "")
    if not commit:
        import subprocess
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True)
        commit = r.stdout.strip() if r.returncode == 0 else "unknown (no .git on this box and no FWA_COMMIT given)"
    ref_path = os.path.join(ROOT, "profiles", "bench_reference.json")
    ref = json.load(open(ref_path))
    f1, n1 = per_dispatch(os.path.join(d, "a_pmc_fetch_summary.txt"), "k_p1_1m")
    f2, _ = per_dispatch(os.path.join(d, "a_pmc_fetch_summary.txt"), "k_p2_1m")
    w1, _ = per_dispatch(os.path.join(d, "a_pmc_write_summary.txt"), "k_p1_1m")
    w2, _ = per_dispatch(os.path.join(d, "a_pmc_write_summary.txt"), "k_p2_1m")
    checks = {}
    for k in ("k_copy", "k_scale", "k_small32<10"):
        try:
            checks[k] = per_dispatch(os.path.join(d, "a_pmc_fetch_summary.txt"), k)[0]
        except SystemExit:
            pass
    launches = 256   # of each kernel per exec at 2^20 x 4096, group 16
    ref["traffic_bytes_per_exec"] = {"1048576x4096": ((2 * f1 + w1) + (2 * f2 + w2)) * 1024 * launches}
    ref["traffic_source"] = (
        f"profiles/round{rnd}/a_pmc_fetch_summary.txt + a_pmc_write_summary.txt: (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch "
        f"(means over {n1} dispatches each: k_p1_1m {f1:.1f} / {w1:.0f} KB, k_p2_1m {f2:.1f} / {w2:.0f} KB) x the 256 k_p1_1m + 256 "
        f"k_p2_1m dispatches of one exec; separate rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 "
        f"--no-cpu-baseline --spread 0` (tools/run_round_profiles.sh, taken BEFORE the bench line of the same job); the gfx950 "
        f"FETCH x2 rule checked in the same run on " + ", ".join(f"{k} ({v:.6g} KB)" for k, v in checks.items()) +
        " (k_copy reads 8 GiB, k_scale 16 GiB, k_small32<10> 32 GiB: FETCH_SIZE = 1/2 of the bytes); WRITE_SIZE exact on all of "
        "them.  TCC counters sit on the L2<->fabric path and include Infinity-Cache hits: this is HBM + ring traffic, 2.02x the "
        "algorithmic bytes by design.")
    u1, c1 = mean_us(os.path.join(d, "a_trace_streams1_summary.txt"), "k_p1_1m")
    u2, _ = mean_us(os.path.join(d, "a_trace_streams1_summary.txt"), "k_p2_1m")
    t1, _ = mean_us(os.path.join(d, "a_trace_default_summary.txt"), "k_p1_1m")
    t2, _ = mean_us(os.path.join(d, "a_trace_default_summary.txt"), "k_p2_1m")
    ref["isolated_kernel_us"] = {
        "k_p1_1m": u1, "k_p2_1m": u2, "transforms_per_launch": 16,
        "source": f"profiles/round{rnd}/a_kernel_stats_streams1_isolated.csv / a_trace_streams1_summary.txt (bench.py --streams 1: one "
                  f"chain, no second launch in flight; {c1} launches each); with two chains the same launches take {t1:.1f} / {t2:.1f} us "
                  f"each while two are in flight (a_trace_default_summary.txt)"}
    ref["taken_on"] = {"commit": commit, "round": int(rnd), "kernel_sources": list(KERNEL_SOURCES),
                       "kernel_source_sha256": kernel_source_sha256()}
    ref["_comment"] = ("Figures bench.py cannot measure inside its timed run, each with its source; refreshed by "
                       "tools/make_bench_reference.py inside the round's profile job, before the bench line is taken.")
    json.dump(ref, open(ref_path, "w"), indent=1)
    print("bench_reference.json refreshed:", ref["traffic_bytes_per_exec"], u1, u2)


if __name__ == "__main__":
    main()
