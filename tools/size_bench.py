#!/usr/bin/env python3
"""tools/size_bench.py -- throughput of Forward.proc for every power of two (total 2^28 samples per size,
C2/C5 shapes with batch 1 as well).  One JSON line per size.  Not part of the product path."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


WARM_MS = 60.0


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--lg-min", type=int, default=1)
    ap.add_argument("--lg-max", type=int, default=24)
    ap.add_argument("--total-lg", type=int, default=28, help="log2 of the samples per size (28 = 2 GiB, 32 = C3's 32 GiB)")
    ap.add_argument("--no-latency-shapes", action="store_true")
    ap.add_argument("--set", default="", help="plan setting applied to every size (tools/sweep.py syntax)")
    args = ap.parse_args()
    from sweep import parse_setting
    kv = parse_setting(args.set)
    dev, queue = fw.prepare_gpu(0)
    enc = dev.create_command_encoder()
    total_lg = args.total_lg
    cases = [(lg, 1 << max(0, total_lg - lg)) for lg in range(args.lg_min, args.lg_max + 1)]
    if not args.no_latency_shapes:
        cases += [(20, 1), (24, 1), (10, 1)]
    buf = dev.create_buffer(8 << max(total_lg, args.lg_max))
    for lg, batch in cases:
        n = 1 << lg
        view = dev.wrap_buffer(buf.device_ptr, n * batch * 8)
        plan = fw.Forward(dev, queue, view, n)
        for key, val in kv.items():
            plan.set(key, val)
        reps = 5 if n * batch >= (1 << 31) else (15 if n * batch >= (1 << 24) else 50)
        times = []
        # Warm-up: at least one exec and at least WARM_MS of them.  For some tens of milliseconds after a multi-GiB hipMalloc /
        # hipFree (a plan of odd log2 n allocates its second buffer) kernels on this part run up to 15 % slower, whatever they
        # touch (profiles/round5/probe_small_footprint_drift.jsonl: 0.82 -> 0.70 ms over the first 20 execs at 2 GiB, flat when
        # the same plan is created a second time); at 32 GiB one exec outlasts it, at 2 GiB five timed execs sat inside it.
        warmed, r = 0.0, 0
        while r < reps + 1:
            dev.fill_synthetic(view, n, scale=2.0 ** -20, encoder=enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            plan.proc(enc)
            b.record(enc)
            ms = a.elapsed_ms(b)
            if warmed < WARM_MS:
                warmed += ms
                if r == 0:
                    r = 1
                continue
            times.append(ms)
            r += 1
        ms = sorted(times)[len(times) // 2]
        print(json.dumps({"lg_n": lg, "batch": batch, "footprint_GiB": round(n * batch * 8 / 2 ** 30, 3), "path": plan.get("path"), "factors": plan.get("factors"), "launches": plan.get("launches_per_exec"),
                          "ms": round(ms, 4), "ms_all": [round(t, 3) for t in times], "Gsamples_s": round(n * batch / ms / 1e6, 2),
                          "roofline_frac": round(16 * n * batch / (ms * 1e-3) / 8e12, 4)}), flush=True)
        plan.destroy()


if __name__ == "__main__":
    main()
