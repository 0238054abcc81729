#!/usr/bin/env python3
"""Does the distance between the source and the destination of an out-of-place streaming pass matter?  The calibration copy
(fwa_calib_copy: the shape of the one-launch kernels) of 8 GiB from `base` to `base + 16 GiB + delta` for a sweep of
`delta`, interleaved, median of `--reps`; then the same for n = 2048 transforms (odd log2 n: out of place by the reference's
result rule) through a plan whose second buffer is a view at that distance (Onlyinverse takes a caller-supplied second buffer).
One JSON line per delta."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_wgpu_amd as fw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--gib", type=int, default=8)
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0)
    enc = dev.create_command_encoder()
    nb = args.gib << 30
    arena = dev.create_buffer(2 * nb + (64 << 20) + (16 << 30))
    src = dev.wrap_buffer(arena.device_ptr, nb)
    deltas = [0, 256, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 65536, 2 << 20, 3 << 20, (4 << 20) + 4096, 16 << 20, (32 << 20) + 131072]
    times = {d: [] for d in deltas}
    fft = {d: [] for d in deltas}
    n = 2048
    dsts = {d: dev.wrap_buffer(arena.device_ptr + (16 << 30) + nb - (16 << 30) + d if False else arena.device_ptr + nb + (32 << 20) + d, nb) for d in deltas}
    plans = {d: fw.Onlyinverse(dev, queue, src, dsts[d], n) for d in deltas}
    for r in range(args.reps + 1):
        for d in deltas:
            dev.fill_synthetic(src, n, scale=2.0 ** -20, encoder=enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            dev.calib_copy(dsts[d], src, nb, encoder=enc)
            b.record(enc)
            ms = a.elapsed_ms(b)
            a2, b2 = fw.Event(dev), fw.Event(dev)
            a2.record(enc)
            plans[d].proc(enc)
            b2.record(enc)
            ms2 = a2.elapsed_ms(b2)
            if r:
                times[d].append(ms)
                fft[d].append(ms2)
    for d in deltas:
        t = sorted(times[d])[len(times[d]) // 2]
        t2 = sorted(fft[d])[len(fft[d]) // 2]
        print(json.dumps({"delta_bytes": d, "copy_GBps": round(2 * nb / (t * 1e-3) / 1e9, 1), "copy_frac": round(2 * nb / (t * 1e-3) / 8e12, 4),
                          "fft2048_frac": round(2 * nb / (t2 * 1e-3) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
