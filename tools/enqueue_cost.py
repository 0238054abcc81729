#!/usr/bin/env python3
"""Host cost of enqueueing one exec, and of ONE process driving several devices.

    python tools/enqueue_cost.py > profiles/round5/enqueue_cost.jsonl

Leg 1: one context, the C3 shape (N = 2^20 x 4096, 32 GiB): wall time for `fwa_plan_exec` (through `Forward.proc`) to
RETURN -- 512 kernel launches, 2 x 256 event record / wait pairs, nothing waited for -- on an idle queue.
Leg 2: `ShardedBatch.proc()` over 8 contexts on device 0 (2^20 x 64 and x 512 per context), enqueued serially and from
one thread per shard; leg 3 the same from a compiled C++ host (tools/enqueue_cost.cpp).  One GPU suffices for the host
side: a launch costs the host the same whichever device it goes to.  The shards share the GPU here, so the "done"
columns are NOT a multi-GPU figure; the "returns" columns are the point: if 8 x the per-device host time approaches one
device's GPU time (21 ms at C3), serial enqueue makes an 8-GPU process host-bound.
"""
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_wgpu_amd as fw  # noqa: E402

N = 1 << 20


def leg_one_context(batch=4096, reps=8):
    dev, queue = fw.prepare_gpu(0)
    buf = dev.create_buffer(N * batch * 8)
    plan = fw.Forward(dev, queue, buf, N)
    enc = dev.create_command_encoder()
    ret, done = [], []
    for r in range(reps + 1):
        dev.fill_synthetic(buf, N, scale=2.0 ** -40, encoder=enc)
        enc.synchronize()
        t0 = time.perf_counter()
        plan.proc(enc)
        t1 = time.perf_counter()
        enc.synchronize()
        t2 = time.perf_counter()
        if r:
            ret.append((t1 - t0) * 1e3)
            done.append((t2 - t0) * 1e3)
    launches = plan.get("launches_per_exec")
    print(json.dumps({"host": "python", "what": "Forward.proc (fwa_plan_exec), one context", "fft_len": N, "batch": batch,
                      "launches": launches, "chains": plan.get("streams"), "returns_ms_median": statistics.median(ret),
                      "returns_ms_min": min(ret), "returns_us_per_launch": statistics.median(ret) * 1e3 / launches,
                      "done_ms_median": statistics.median(done), "reps": reps}), flush=True)
    plan.destroy()
    buf.destroy()
    enc.destroy()
    dev.destroy()


def leg_sharded(shards, per, reps=5):
    for threads in (False, True):
        sb = fw.ShardedBatch(fw.Forward, N, per * shards, ordinals=[0] * shards, threads=threads)
        ret, done = [], []
        for r in range(reps + 1):
            for d, b, e, (a, z) in zip(sb.devices, sb.buffers, sb.encoders, sb.slabs):
                d.fill_synthetic(b, N, first_transform=a, scale=2.0 ** -40, encoder=e)
            sb.poll()
            t0 = time.perf_counter()
            sb.proc()
            t1 = time.perf_counter()
            sb.poll()
            t2 = time.perf_counter()
            if r:
                ret.append((t1 - t0) * 1e3)
                done.append((t2 - t0) * 1e3)
        print(json.dumps({"host": "python", "what": "ShardedBatch.proc", "enqueue": "threaded" if threads else "serial",
                          "shards": shards, "fft_len": N, "transforms_per_shard": per,
                          "launches_per_shard": sb.plans[0].get("launches_per_exec"),
                          "returns_ms_median": statistics.median(ret), "returns_ms_min": min(ret),
                          "done_ms_median": statistics.median(done), "reps": reps}), flush=True)
        sb.destroy()


def leg_cpp(shards, per):
    exe = os.path.join("/tmp", "enqueue_cost")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "enqueue_cost.cpp"),
                           "-L" + os.path.join(ROOT, "fft_wgpu_amd"), "-lfft_wgpu_amd", "-pthread", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "fft_wgpu_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, str(shards), str(per), "5"], env=env, capture_output=True, text=True, check=True).stdout
    sys.stdout.write(out)
    sys.stdout.flush()


if __name__ == "__main__":
    leg_one_context()
    for per in (64, 512):
        leg_sharded(8, per)
    for per in (64, 512):
        leg_cpp(8, per)
