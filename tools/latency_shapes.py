#!/usr/bin/env python3
"""tools/latency_shapes.py -- the batch-1 (latency) shapes of BASELINE.json, one JSON line per shape.

C2 (2^20 x 1), C5 (2^24 x 1), 2^16 x 1, 2^18 x 1, 1024 x 1 (C1's shape), `--reps` executions each, two clocks:

  host_us   HIP events around `proc` on an idle stream: what a caller sees -- the host's launch calls are inside
            (event a retires before the first kernel has been enqueued);
  queued_us the same events with a blocker (a 256-MiB calibration copy) enqueued first, so that every launch of the exec is
            already in the queue when the device reaches event a: device-side time of the launches alone.

`--tree DIR` imports fft_wgpu_amd from another checkout (A/B against an older round: tools/variants/<name>).
Not part of the product path.
"""
import argparse
import json
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--label", default="")
    ap.add_argument("--shapes", default="20x1,24x1,16x1,18x1,10x1")
    args = ap.parse_args()
    sys.path.insert(0, os.path.abspath(args.tree))
    import fft_wgpu_amd as fw
    dev, queue = fw.prepare_gpu(0)
    enc = dev.create_command_encoder()
    blk = 256 << 20
    blocker = dev.create_buffer(2 * blk)
    bsrc, bdst = dev.wrap_buffer(blocker.device_ptr, blk), dev.wrap_buffer(blocker.device_ptr + blk, blk)
    for shape in args.shapes.split(","):
        lg, batch = (int(t) for t in shape.split("x"))
        n = 1 << lg
        buf = dev.create_buffer(n * batch * 8)
        plan = fw.Forward(dev, queue, buf, n)
        res = {}
        for mode in ("host", "queued"):
            times = []
            for r in range(args.reps + 3):
                dev.fill_synthetic(buf, n, scale=2.0 ** -20, encoder=enc)
                enc.synchronize()
                a, b = fw.Event(dev), fw.Event(dev)
                if mode == "queued":
                    dev.calib_copy(bdst, bsrc, blk, encoder=enc)
                a.record(enc)
                plan.proc(enc)
                b.record(enc)
                ms = a.elapsed_ms(b)
                if r >= 3:
                    times.append(ms * 1e3)
            times.sort()
            res[mode] = times
        line = {"label": args.label, "lg_n": lg, "batch": batch, "path": plan.get("path"), "factors": plan.get("factors"),
                "launches": plan.get("launches_per_exec"), "reps": args.reps}
        for mode, t in res.items():
            line[mode + "_us"] = round(t[len(t) // 2], 2)
            line[mode + "_us_min"] = round(t[0], 2)
            line[mode + "_us_p90"] = round(t[int(len(t) * 0.9)], 2)
        print(json.dumps(line), flush=True)
        plan.destroy()
        buf.destroy()


if __name__ == "__main__":
    main()
