#!/usr/bin/env python3
"""tools/reference_loop.py -- the reference's own "benchmark" (src/examples/basic.rs:72-129): N = 512,
batch 2500, all-ones input; every iteration uploads the batch, runs Forward.proc, copies the result to a
staging buffer and reads it back.  Measured two ways through the C ABI:
  serial    -- the reference's sequence, one blocking iteration after another;
  pipelined -- fft_wgpu_amd.HostPipeline: pinned staging, upload + transform on one stream, read-back on a second,
               double or triple buffering.
Reports iterations/s and the PCIe-inclusive sample rate (this is NOT bench.py's `value`)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--batch", type=int, default=2500)
    ap.add_argument("--iters", type=int, default=1000)
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0)
    n, batch = args.n, args.batch
    count = n * batch
    nbytes = count * 8

    # ---- serial, pageable host memory: exactly the reference's call sequence
    data = np.ones(count, dtype=np.complex64)
    src = dev.create_buffer(nbytes)
    staging = dev.create_buffer(nbytes)
    plan = fw.Forward(dev, queue, src, n)
    enc = dev.create_command_encoder()
    for it in range(3 + args.iters):
        if it == 3:
            t0 = time.perf_counter()
        queue.write_buffer(src, 0, data, encoder=enc)                  # basic.rs:73
        out = plan.proc(enc)                                           # :79
        enc.copy_buffer_to_buffer(out, 0, staging, 0, nbytes)          # :84-90
        ans = staging.map_read(stream=enc)                             # :92-122
    dt = time.perf_counter() - t0
    assert abs(ans[0] - n) < 1e-3 and abs(ans[1]) < 1e-5
    print(json.dumps({"what": "reference loop, serial (pageable)", "n": n, "batch": batch, "iters": args.iters,
                      "iters_per_s": args.iters / dt, "Gsamples_s_pcie_inclusive": count * args.iters / dt / 1e9,
                      "host_link_GBps_each_way": nbytes * args.iters / dt / 1e9}), flush=True)

    # ---- pipelined: pinned staging, double buffering, three streams, one event per slot and stage
    for slots in (2, 3):
        pipe = fw.HostPipeline(dev, queue, lambda d, q, b: fw.Forward(d, q, b, n), count, slots=slots)
        for h in pipe.hin:
            h[:] = 1
        for it in range(2 * slots + args.iters):
            if it == 2 * slots:
                pipe.drain()
                t0 = time.perf_counter()
            pipe.submit()
        pipe.drain()
        dt = time.perf_counter() - t0
        for s in range(slots):
            r = pipe.result(s)
            assert abs(r[0] - n) < 1e-3 and abs(r[n] - n) < 1e-3 and abs(r[1]) < 1e-5
        print(json.dumps({"what": f"reference loop, pipelined (fft_wgpu_amd.HostPipeline: pinned staging, 2 streams, {slots} slots)", "n": n,
                          "batch": batch, "iters": args.iters, "iters_per_s": args.iters / dt,
                          "Gsamples_s_pcie_inclusive": count * args.iters / dt / 1e9,
                          "host_link_GBps_each_way": nbytes * args.iters / dt / 1e9}), flush=True)
        del pipe


if __name__ == "__main__":
    main()
