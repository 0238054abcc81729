#!/usr/bin/env python3
"""tools/sweep.py -- GPU-side tuning sweep (not part of the product path or the tests).

Measures (HIP events, median of a few repetitions):
  * float4 copy rate vs footprint (what the Infinity Cache serves vs HBM),
  * the N=2^20 plan at several (group, streams) settings.
Writes one JSON line per measurement to stdout.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--settings", type=str, default="8x2", help="two-launch path: GROUPxSTREAMS,...")
    ap.add_argument("--mix", type=str, default="", help="mixed-launch two-pass path: GROUPxSTREAMSxDBGxPOLICY,...")
    ap.add_argument("--fused", type=str, default="4x512,2x512,3x512,6x512,8x512,12x512,4x256,4x384",
                    help="fused path: DEPTHxWORKGROUPS,...")
    ap.add_argument("--copy", action="store_true")
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0)
    n = 1 << 20
    nbytes = n * args.batch * 8
    buf = dev.create_buffer(nbytes)
    enc = dev.create_command_encoder()
    dev.fill_synthetic(buf, n, scale=2.0 ** -40, encoder=enc)
    enc.synchronize()

    if args.copy:
        for mib in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
            half = mib << 20
            if 2 * half > nbytes:
                break
            s = dev.wrap_buffer(buf.device_ptr, half)
            d = dev.wrap_buffer(buf.device_ptr + half, half)
            iters = max(3, min(200, (8 << 30) // half))
            for _ in range(2):
                dev.calib_copy(d, s, half, encoder=enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            for _ in range(iters):
                dev.calib_copy(d, s, half, encoder=enc)
            b.record(enc)
            ms = a.elapsed_ms(b) / iters
            print(json.dumps({"what": "copy", "footprint_MiB": 2 * mib, "us": ms * 1e3,
                              "GBps_rw": 2 * half / (ms * 1e-3) / 1e9}), flush=True)
        dev.fill_synthetic(buf, n, scale=2.0 ** -40, encoder=enc)
        enc.synchronize()

    runs = ([("two", s) for s in args.settings.split(",") if s] + [("mix", s) for s in args.mix.split(",") if s]
            + [("fused", s) for s in args.fused.split(",") if s])
    for kind, s in runs:
        parts = [int(v) for v in s.split("x")]
        g, ns = parts[0], parts[1]
        dbg = parts[2] if len(parts) > 2 else 0
        pol = parts[3] if len(parts) > 3 else 0
        plan = fw.Forward(dev, queue, buf, n)
        if kind in ("two", "mix"):
            plan.set("path", 1)
            plan.set("mix", 1 if kind == "mix" else 0)
            plan.set("group", g)
            plan.set("streams", ns)
        else:
            plan.set("depth", g)
            plan.set("wgs", ns)
            if dbg:
                plan.set("dbg", dbg)
        plan.set("policy", pol)
        dev.fill_synthetic(buf, n, scale=2.0 ** -40, encoder=enc)
        plan.proc(enc)
        enc.synchronize()
        times = []
        for _ in range(args.reps):
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            plan.proc(enc)
            b.record(enc)
            times.append(a.elapsed_ms(b))
        ms = med(times)
        err = plan.get("device_error")
        print(json.dumps({"what": "fft1m_" + kind, "a": g, "b": ns, "dbg": dbg, "policy": pol, "device_error": err, "batch": args.batch, "ms": ms,
                          "ms_all": times, "Gsamples_s": n * args.batch / (ms * 1e-3) / 1e9,
                          "roofline_frac": 16 * n * args.batch / (ms * 1e-3) / 8e12}), flush=True)
        plan.destroy()


if __name__ == "__main__":
    main()
