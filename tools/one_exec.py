#!/usr/bin/env python3
"""tools/one_exec.py -- run a few Forward.proc calls of one plan setting (for rocprofv3 kernel traces / PMC passes).
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/x -- python3 tools/one_exec.py --lg 16 --batch 4096 --set path=8"""
import argparse
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
    ap.add_argument("--lab", action="store_true", help="load the laboratory build (paths 5 / 8, tile_w = 32, small_reg != 1)")
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0, lab=args.lab)
    n = 1 << args.lg
    buf = dev.create_buffer(n * args.batch * 8)
    enc = dev.create_command_encoder()
    plan = fw.Forward(dev, queue, buf, n)
    kv = parse_setting(args.set)
    for key in PLAN_KEYS:
        if key in kv:
            plan.set(key, kv[key])
    for _ in range(args.execs):
        dev.fill_synthetic(buf, n, scale=2.0 ** -40, encoder=enc)
        plan.proc(enc)
    enc.synchronize()
    print("done", plan.get("path"), plan.get("launches_per_exec"))


if __name__ == "__main__":
    main()
