#!/usr/bin/env python3
"""tools/pipe_long_probe.py [iters] [slots] -- HostPipeline rate per block of 100 iterations over a long run (does the rate hold?)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100   # drain + clock every this many iterations
dev, queue = fw.prepare_gpu(0)
n, batch = 512, 2500
count = n * batch
pipe = fw.HostPipeline(dev, queue, lambda d, q, b: fw.Forward(d, q, b, n), count, slots=slots)
for h in pipe.hin:
    h[:] = 1
rates = []
for it in range(iters):
    if it % every == 0:
        pipe.drain()
        t = time.perf_counter()
        if it:
            rates.append(round(every / (t - t0), 0))
        t0 = t
    pipe.submit()
pipe.drain()
rates.append(round(every / (time.perf_counter() - t0), 0))
print(json.dumps({"slots": slots, "iters": iters, "drain_every": every, "iters_per_s_per_block": rates}), flush=True)
