#!/usr/bin/env python3
"""tools/sweep.py -- GPU-side tuning sweep (not part of the product path or the tests).

Times Forward.proc (HIP events on the launch stream, median of --reps) for a list of plan settings, all in ONE
process on ONE device with the variants interleaved round-robin (cdna_hip_programming.md rule 24).
A setting is "key=value,key=value" over the plan's tunables (group, streams, tile_w, cw, factors); factors may be
written as 9.9 or 6.6.6.  One JSON line per setting.

  python tools/sweep.py --lg 20 --batch 4096 --set "group=16" --set "group=8,streams=4" --set "xcd_swizzle=1"
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402
from fft_wgpu_amd.processor import PLAN_KEYS  # noqa: E402


def parse_setting(s):
    out = {}
    for kv in [p for p in s.split(",") if p]:
        k, v = kv.split("=")
        if k == "factors":
            f = [int(t) for t in v.split(".")] + [0]
            out[k] = f[0] | (f[1] << 8) | (f[2] << 16)
        else:
            out[k] = int(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lg", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--set", action="append", default=[], help="one plan setting (repeatable); '' = defaults")
    ap.add_argument("--copy", action="store_true", help="float4 copy rate vs footprint first")
    ap.add_argument("--lab", action="store_true", help="load the laboratory build (path 5, small_reg = 2 / 3, ring_rotate)")
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0, lab=args.lab)
    n = 1 << args.lg
    nbytes = n * args.batch * 8
    buf = dev.create_buffer(nbytes)
    enc = dev.create_command_encoder()
    scale = 2.0 ** -40

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

    settings = args.set or [""]
    plans = []
    for s in settings:
        plan = fw.Forward(dev, queue, buf, n)
        kv = parse_setting(s)
        for key in PLAN_KEYS:
            if key in kv:
                plan.set(key, kv[key])
        plans.append((s, plan, []))
    # one regeneration of the input per round: log2(n) <= 30 bits of growth per exec, 2^-40 input scale and at most
    # 3 execs between refills keep fp32 finite
    for r in range(args.reps + 1):
        for i, (s, plan, times) in enumerate(plans):
            if i % 3 == 0:
                dev.fill_synthetic(buf, n, scale=scale, encoder=enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            plan.proc(enc)
            b.record(enc)
            ms = a.elapsed_ms(b)
            if r:
                times.append(ms)
    for s, plan, times in plans:
        ms = sorted(times)[len(times) // 2]
        print(json.dumps({"what": "fft", "lg_n": args.lg, "batch": args.batch, "setting": s or "(default)",
                          "path": plan.get("path"), "factors": plan.get("factors"), "group": plan.get("group"),
                          "streams": plan.get("streams"), "tile_w": plan.get("tile_w"), "xcd_swizzle": plan.get("xcd_swizzle"),
                          "ms": round(ms, 4), "ms_min": round(min(times), 4), "ms_all": [round(t, 4) for t in times],
                          "Gsamples_s": round(n * args.batch / (ms * 1e-3) / 1e9, 2),
                          "roofline_frac": round(16 * n * args.batch / (ms * 1e-3) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
