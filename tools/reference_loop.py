#!/usr/bin/env python3
"""tools/reference_loop.py -- the reference's own "benchmark" (src/examples/basic.rs:72-129): N = 512,
batch 2500, all-ones input; every iteration uploads the batch, runs Forward.proc, copies the result to a
staging buffer and reads it back.  Measured two ways through the C ABI:
  serial    -- the reference's sequence, one blocking iteration after another;
  pipelined -- pinned staging + three streams (upload / transform / download) with double buffering.
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

    # ---- pipelined: pinned staging, double buffering, three streams
    up, ex, down = (dev.create_command_encoder() for _ in range(3))
    hin = [dev.pinned_array(count) for _ in range(2)]
    hout = [dev.pinned_array(count) for _ in range(2)]
    for h in hin:
        h[:] = 1
    bufs = [dev.create_buffer(nbytes) for _ in range(2)]
    outs = [dev.create_buffer(nbytes) for _ in range(2)]
    plans = [fw.Forward(dev, queue, b, n) for b in bufs]
    for it in range(4 + args.iters):
        if it == 4:
            dev.poll(up); dev.poll(ex); dev.poll(down)
            t0 = time.perf_counter()
        s = it & 1
        up.wait_for(ex)                                                # slot s was consumed two iterations ago
        queue.write_buffer(bufs[s], 0, hin[s], encoder=up)
        ex.wait_for(up)
        ex.wait_for(down)                                              # outs[s] fully read back before reuse
        out = plans[s].proc(ex)
        ex.copy_buffer_to_buffer(out, 0, outs[s], 0, nbytes)
        down.wait_for(ex)
        dev.download_async(hout[s], outs[s], down)
    dev.poll(up); dev.poll(ex); dev.poll(down)
    dt = time.perf_counter() - t0
    assert abs(hout[0][0] - n) < 1e-3 and abs(hout[1][n] - n) < 1e-3
    print(json.dumps({"what": "reference loop, pipelined (pinned, 3 streams)", "n": n, "batch": batch,
                      "iters": args.iters, "iters_per_s": args.iters / dt,
                      "Gsamples_s_pcie_inclusive": count * args.iters / dt / 1e9,
                      "host_link_GBps_each_way": nbytes * args.iters / dt / 1e9}), flush=True)


if __name__ == "__main__":
    main()
