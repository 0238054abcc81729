#!/usr/bin/env python3
"""tools/kinds_bench.py -- the four plan kinds at one size (HIP events, median): Forward, Inverse (fused 1/n),
Onlyinverse, Normalize.  One JSON line per (size, kind)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def main():
    dev, queue = fw.prepare_gpu(0)
    enc = dev.create_command_encoder()
    for lg, batch in ((9, 1 << 19), (12, 1 << 16), (16, 4096), (20, 1024), (24, 16)):
        n = 1 << lg
        a = dev.create_buffer(n * batch * 8)
        b = dev.create_buffer(n * batch * 8)
        plans = {"Forward": fw.Forward(dev, queue, a, n), "Inverse": fw.Inverse(dev, queue, a, n),
                 "Onlyinverse": fw.Onlyinverse(dev, queue, a, b, n), "Normalize": fw.Normalize(dev, queue, a, b, n)}
        for kind, plan in plans.items():
            times = []
            warmed = 0.0      # >= 60 ms of untimed execs first: the transient after multi-GiB allocations (tools/size_bench.py)
            while len(times) < 9:
                dev.fill_synthetic(a, n, scale=2.0 ** -20, encoder=enc)
                e0, e1 = fw.Event(dev), fw.Event(dev)
                e0.record(enc)
                plan.proc(enc)
                e1.record(enc)
                ms = e0.elapsed_ms(e1)
                if warmed < 60.0:
                    warmed += ms
                else:
                    times.append(ms)
            ms = sorted(times)[len(times) // 2]
            print(json.dumps({"lg_n": lg, "batch": batch, "kind": kind, "ms": round(ms, 4),
                              "Gsamples_s": round(n * batch / ms / 1e6, 1),
                              "roofline_frac": round(16 * n * batch / (ms * 1e-3) / 8e12, 4)}), flush=True)
        for p in plans.values():
            p.destroy()
        a.destroy()
        b.destroy()


if __name__ == "__main__":
    main()
