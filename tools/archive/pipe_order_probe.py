#!/usr/bin/env python3
"""tools/pipe_order_probe.py MODE -- why does tools/reference_loop.py's pipelined leg read 21 GB/s each way when the same
HostPipeline alone reads 44?  One mode per process:
  alone      HostPipeline only
  after      a serial encoder is created and used first (reference_loop.py's order), kept alive
  destroyed  the serial encoder is destroyed before the pipeline is built
  idle       a serial encoder is created but never used
  nocheck    as `after`, with the stream-overlap check off (fwa_ctx_set_i64 chain_check = 0)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402

mode = sys.argv[1]
dev, queue = fw.prepare_gpu(0)
n, batch, iters = 512, 2500, 400
count = n * batch
if mode == "nocheck":
    dev.set("chain_check", 0)
if mode != "alone":
    enc = dev.create_command_encoder()
    if mode != "idle":
        data = np.ones(count, dtype=np.complex64)
        src = dev.create_buffer(count * 8)
        staging = dev.create_buffer(count * 8)
        plan = fw.Forward(dev, queue, src, n)
        for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
            queue.write_buffer(src, 0, data, encoder=enc)
            out = plan.proc(enc)
            enc.copy_buffer_to_buffer(out, 0, staging, 0, count * 8)
            staging.map_read(stream=enc)
    if mode == "destroyed":
        enc.destroy()
pipe = fw.HostPipeline(dev, queue, lambda d, q, b: fw.Forward(d, q, b, n), count, slots=2)
for h in pipe.hin:
    h[:] = 1
for it in range(4 + iters):
    if it == 4:
        pipe.drain()
        t0 = time.perf_counter()
    pipe.submit()
pipe.drain()
dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "iters_per_s": round(iters / dt, 1), "GBps_each_way": round(count * 8 * iters / dt / 1e9, 2), **{k: v for k, v in dev.stats().items() if k.startswith("chain")}}), flush=True)
