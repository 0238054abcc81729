#!/usr/bin/env python3
"""tools/plan_mode_probe.py -- does the time of an exec depend on the plan INSTANCE (its internal streams / ring)?
Creates the same plan N times in one process (destroying or keeping the previous ones) and times each instance."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lg", type=int, default=19)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--plans", type=int, default=12)
    ap.add_argument("--keep", action="store_true", help="keep earlier plans alive")
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0)
    n = 1 << args.lg
    buf = dev.create_buffer(n * args.batch * 8)
    enc = dev.create_command_encoder()
    kept = []
    out = []
    for i in range(args.plans):
        plan = fw.Forward(dev, queue, buf, n)
        times = []
        for r in range(4):
            dev.fill_synthetic(buf, n, scale=2.0 ** -40, encoder=enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            plan.proc(enc)
            b.record(enc)
            if r:
                times.append(a.elapsed_ms(b))
        out.append(round(sorted(times)[1], 3))
        if args.keep:
            kept.append(plan)
        else:
            plan.destroy()
    print(json.dumps({"lg_n": args.lg, "batch": args.batch, "keep": args.keep, "hwq": os.environ.get("GPU_MAX_HW_QUEUES", ""),
                      "ms_per_plan_instance": out}), flush=True)


if __name__ == "__main__":
    main()
