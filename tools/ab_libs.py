#!/usr/bin/env python3
"""A/B of builds of the SAME sources with different compile-time defaults (tools/build_variant.sh), in one process,
interleaved: every library gets its own context on device 0.

    python tools/ab_libs.py --libs base=build/variants/base/libfft_wgpu_amd.so,pf=build/variants/pf/libfft_wgpu_amd.so \
        --shapes 16,17,18,19,21,22,23,20x1,24x1 [--total-lg 32] [--rounds 4]

Shape `L` = 2^L-point transforms filling 2^total-lg samples (throughput: HIP events around one exec, median); `LxB` = a batch of B
(latency shapes: `queued_us`, HIP events with a 256-MiB copy enqueued first so that every launch is queued when the clock starts,
as tools/latency_shapes.py).  First a bit-comparison of the libraries' results on a small batch (the variants only move
look-ups: identical bits expected; reported, not assumed).  One JSON line per (shape, library).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_wgpu_amd as fw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--shapes", required=True)
    ap.add_argument("--total-lg", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kind", default="Forward")
    args = ap.parse_args()
    libs = [tuple(t.split("=", 1)) for t in args.libs.split(",")]
    ctx = {}
    for name, path in libs:
        dev = fw.Device(0, lab=os.path.abspath(path))
        ctx[name] = (dev, fw.Queue(dev), dev.create_command_encoder())
    blk = 256 << 20
    two = args.kind in ("Onlyinverse", "Normalize")     # plan kinds with a caller-supplied second buffer
    for shape in args.shapes.split(","):
        lat = "x" in shape
        lg, batch = (int(t) for t in shape.split("x")) if lat else (int(shape), 0)
        n = 1 << lg
        if not lat:
            batch = 1 << max(0, args.total_lg - lg)
        # identical bits on a small batch
        small = min(batch, max(1, (1 << 22) >> lg) + 1)
        d0, _, e0 = ctx[libs[0][0]]
        g = d0.create_buffer(8 * n * small)
        d0.fill_synthetic(g, n, first_transform=5, encoder=e0)
        x = g.map_read(stream=e0)
        g.destroy()
        outs = {}
        for name, (dev, queue, enc) in ctx.items():
            b = dev.create_buffer(x.nbytes)
            queue.write_buffer(b, 0, x)
            b2 = dev.create_buffer(x.nbytes) if two else None
            if two and n.bit_length() % 2 == 0:   # odd log2 n: Normalize reads the SECOND buffer (processor.rs:433-439)
                queue.write_buffer(b2, 0, x)
            p = getattr(fw, args.kind)(dev, queue, b, b2, n) if two else getattr(fw, args.kind)(dev, queue, b, n)
            outs[name] = p.proc(enc).map_read(stream=enc)
            p.destroy()
            b.destroy()
            if b2 is not None:
                b2.destroy()
        ref = outs[libs[0][0]]
        same = {name: bool(np.array_equal(ref.view(np.uint32), y.view(np.uint32))) for name, y in outs.items()}
        worst = {name: float(np.abs(y - ref).max() / np.abs(ref).max()) for name, y in outs.items()}
        plans, bufs, blockers, seconds = {}, {}, {}, {}
        for name, (dev, queue, enc) in ctx.items():
            bufs[name] = dev.create_buffer(8 * n * batch)
            seconds[name] = dev.create_buffer(8 * n * batch) if two else None
            plans[name] = (getattr(fw, args.kind)(dev, queue, bufs[name], seconds[name], n) if two
                           else getattr(fw, args.kind)(dev, queue, bufs[name], n))
            if lat:
                bl = dev.create_buffer(2 * blk)
                blockers[name] = (bl, dev.wrap_buffer(bl.device_ptr, blk), dev.wrap_buffer(bl.device_ptr + blk, blk))
        t = {name: [] for name in ctx}
        reps = 12 if lat else args.reps
        if not lat:   # >= 60 ms of untimed execs per library: the transient after multi-GiB allocations (tools/size_bench.py)
            for name, (dev, queue, enc) in ctx.items():
                warmed = 0.0
                while warmed < 60.0:
                    dev.fill_synthetic(bufs[name], n, scale=2.0 ** -20, encoder=enc)
                    a, b = fw.Event(dev), fw.Event(dev)
                    a.record(enc)
                    plans[name].proc(enc)
                    b.record(enc)
                    warmed += a.elapsed_ms(b)
        for rd in range(args.rounds + 1):
            for name, (dev, queue, enc) in ctx.items():
                for r in range(reps):
                    dev.fill_synthetic(bufs[name], n, scale=2.0 ** -20, encoder=enc)
                    if lat:
                        enc.synchronize()
                        dev.calib_copy(blockers[name][2], blockers[name][1], blk, encoder=enc)
                    a, b = fw.Event(dev), fw.Event(dev)
                    a.record(enc)
                    plans[name].proc(enc)
                    b.record(enc)
                    ms = a.elapsed_ms(b)
                    if rd:
                        t[name].append(ms)
        for name in ctx:
            v = sorted(t[name])
            med = v[len(v) // 2]
            line = {"shape": shape, "lg_n": lg, "batch": batch, "lib": name, "kind": args.kind, "path": plans[name].get("path"),
                    "factors": plans[name].get("factors"), "launches": plans[name].get("launches_per_exec"),
                    "bits_equal_to_first_lib": same[name], "max_rel_diff_to_first_lib": worst[name], "samples": len(v)}
            if lat:
                line.update({"queued_us": round(med * 1e3, 2), "queued_us_min": round(v[0] * 1e3, 2)})
            else:
                line.update({"ms_median": round(med, 4), "ms_min": round(v[0], 4),
                             "roofline_frac": round(16 * n * batch / (med * 1e-3) / 8e12, 4)})
            print(json.dumps(line), flush=True)
        for name in ctx:
            plans[name].destroy()
            bufs[name].destroy()
            if seconds[name] is not None:
                seconds[name].destroy()
            if lat:
                for h in blockers[name][1:] + blockers[name][:1]:
                    h.destroy()


if __name__ == "__main__":
    main()
