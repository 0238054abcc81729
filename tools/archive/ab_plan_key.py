#!/usr/bin/env python3
"""A/B one plan key at one size: parity of every value against numpy's fp64 FFT on small ragged batches, then interleaved
timing at a footprint (default 32 GiB of samples, C3's).  One JSON line per value.

    python tools/ab_plan_key.py --lg 9 --key wave --values 0,1 [--total-lg 32] [--rounds 4]
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
    ap.add_argument("--lg", type=int, required=True)
    ap.add_argument("--key", required=True)
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--total-lg", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n = 1 << args.lg
    values = [int(v) for v in args.values.split(",")]
    dev, queue = fw.prepare_gpu(0)
    enc = dev.create_command_encoder()
    # parity: forward against numpy's fp64 FFT of the device generator's output, inverse round trip, ragged batches
    worst = {v: 0.0 for v in values}
    for batch in (1, 3, 16, 37, 4099 if n <= 4096 else 5):
        g = dev.create_buffer(8 * n * batch)
        dev.fill_synthetic(g, n, first_transform=7, encoder=enc)
        x = g.map_read(stream=enc)
        g.destroy()
        r = np.fft.fft(x.astype(np.complex128).reshape(batch, n), axis=1)
        for v in values:
            b = dev.create_buffer(x.nbytes)
            queue.write_buffer(b, 0, x)
            p = fw.Forward(dev, queue, b, n)
            p.set(args.key, v)
            out = p.proc(enc)
            y = out.map_read(stream=enc).reshape(batch, n)
            err = np.abs(y - r).max(axis=1) / np.abs(r).max(axis=1)
            assert err.max() <= 1e-5, (args.key, v, batch, float(err.max()))
            worst[v] = max(worst[v], float(err.max()))
            q = fw.Inverse(dev, queue, out, n)
            q.set(args.key, v)
            z = q.proc(enc).map_read(stream=enc)
            assert np.abs(z - x).max() <= 1e-5 * np.abs(x).max(), (args.key, v, batch)
            p.destroy(); q.destroy(); b.destroy()
    # timing, interleaved
    batch = 1 << max(0, args.total_lg - args.lg)
    buf = dev.create_buffer(8 * n * batch)
    plans = {}
    for v in values:
        plans[v] = fw.Forward(dev, queue, buf, n)
        plans[v].set(args.key, v)
    ms = {v: [] for v in values}
    for rd in range(args.rounds):
        for v in values:
            for r in range(args.reps + 1):
                dev.fill_synthetic(buf, n, scale=2.0 ** -20, encoder=enc)
                a, b = fw.Event(dev), fw.Event(dev)
                a.record(enc)
                plans[v].proc(enc)
                b.record(enc)
                if r:
                    ms[v].append(a.elapsed_ms(b))
    for v in values:
        t = sorted(ms[v])
        med = t[len(t) // 2]
        print(json.dumps({"lg_n": args.lg, "batch": batch, "key": args.key, "value": v, "ms_median": round(med, 4), "ms_min": round(t[0], 4),
                          "ms_max": round(t[-1], 4), "roofline_frac": round(16 * n * batch / (med * 1e-3) / 8e12, 4),
                          "roofline_frac_best": round(16 * n * batch / (t[0] * 1e-3) / 8e12, 4),
                          "max_rel_err_vs_dft_f64": worst[v], "samples": len(t)}), flush=True)


if __name__ == "__main__":
    main()
