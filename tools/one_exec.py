#!/usr/bin/env python3
"""tools/one_exec.py -- run a few `proc` calls of one plan setting (a rocprofv3 target: kernel traces / PMC passes).
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/x -- python3 tools/one_exec.py --lg 16 --batch 4096
Prints one JSON line: what the plan chose and the HIP-event time of every exec (input regenerated outside the events)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402
from fft_wgpu_amd.processor import PLAN_KEYS  # noqa: E402
from sweep import parse_setting  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lg", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--execs", type=int, default=3)
    ap.add_argument("--set", default="")
    ap.add_argument("--kind", default="forward", choices=["forward", "normalize"])
    ap.add_argument("--lab", action="store_true", help="load the laboratory build (path 5, small_reg = 2 / 3, ring_rotate)")
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0, lab=args.lab)
    n = 1 << args.lg
    enc = dev.create_command_encoder()
    if args.kind == "normalize":       # normalize.wgsl:9-12: buffer1 -> buffer2, 16 B per sample like a transform
        buf = dev.create_buffer(n * args.batch * 8)
        second = dev.create_buffer(n * args.batch * 8)
        plan = fw.Normalize(dev, queue, buf, second, n)
    else:
        buf = dev.create_buffer(n * args.batch * 8)
        plan = fw.Forward(dev, queue, buf, n)
    kv = parse_setting(args.set)
    for key in PLAN_KEYS:
        if key in kv:
            plan.set(key, kv[key])
    ms = []
    for _ in range(args.execs):
        dev.fill_synthetic(buf, n, scale=2.0 ** -40, encoder=enc)
        a, b = fw.Event(dev), fw.Event(dev)
        a.record(enc)
        plan.proc(enc)
        b.record(enc)
        ms.append(a.elapsed_ms(b))
    enc.synchronize()
    print(json.dumps({"one_exec": args.kind, "lg_n": args.lg, "batch": args.batch, "path": plan.get("path"),
                      "factors": plan.get("factors"), "launches_per_exec": plan.get("launches_per_exec"),
                      "streams": plan.get("streams") if plan.get("path") in (1, 7) else 1, "exec_ms": [round(t, 4) for t in ms]}))


if __name__ == "__main__":
    main()
