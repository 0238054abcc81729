#!/usr/bin/env python3
"""tools/fused_c2_ab.py -- VERDICT round 5, item 7: config C2 (one 2^20 transform) as ONE launch (laboratory key "fused" = 1)
against the shipped three k_tile launches.  First parity: both forms on the same inputs, the fused result bit-identical to the
three-launch one and both within 1e-5 of the fp64 DFT (numpy), alternating inputs on ONE plan so that a stale slab line of the
previous exec would show; Forward and Inverse; `device_error` must stay 0.  Then timing, interleaved in one process: HIP events
around one `proc` on an idle stream and behind a blocker copy (tools/latency_shapes.py's two clocks), `--reps` execs each.
One JSON line per form."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=12)
    args = ap.parse_args()
    dev, queue = fw.prepare_gpu(0, lab=True)
    enc = dev.create_command_encoder()
    n = 1 << 20
    rng = np.random.default_rng(7)
    bufs = {k: dev.create_buffer(n * 8) for k in ("three", "fused")}
    plans = {}
    for kind in (fw.Forward, fw.Inverse):
        for form in ("three", "fused"):
            p = kind(dev, queue, bufs[form], n)
            if form == "fused":
                p.set("fused", 1)
            assert p.get("launches_per_exec") == (1 if form == "fused" else 3), p.get("launches_per_exec")
            plans[(kind.__name__, form)] = p
    worst = 0.0
    for r in range(args.rounds):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(1 + r)
        for kind, sign in (("Forward", -1), ("Inverse", +1)):
            got = {}
            for form in ("three", "fused"):
                queue.write_buffer(bufs[form], 0, x, encoder=enc)
                got[form] = plans[(kind, form)].proc(enc).map_read(stream=enc)
            ref = np.fft.fft(x.astype(np.complex128)) if sign < 0 else np.fft.ifft(x.astype(np.complex128))
            err = float(np.abs(got["fused"].astype(np.complex128) - ref).max() / np.abs(ref).max())
            worst = max(worst, err)
            same = np.array_equal(got["fused"].view(np.uint32), got["three"].view(np.uint32))
            if not same or err > 1e-5:
                bad = np.flatnonzero(got["fused"].view(np.uint64) != got["three"].view(np.uint64))
                print(json.dumps({"parity": "FAILED", "round": r, "kind": kind, "bit_identical": bool(same), "max_rel": err,
                                  "mismatching_samples": int(bad.size), "first": [int(b) for b in bad[:8]]}), flush=True)
                sys.exit(1)
    errs = {k: plans[(k, "fused")].get("device_error") for k in ("Forward", "Inverse")}
    print(json.dumps({"parity": "ok", "rounds": args.rounds, "kinds": ["Forward", "Inverse"], "bit_identical_to_three_launches": True,
                      "worst_max_rel_vs_fp64": worst, "device_error": errs}), flush=True)
    assert not any(errs.values())
    # timing
    blk = 256 << 20
    blocker = dev.create_buffer(2 * blk)
    bsrc, bdst = dev.wrap_buffer(blocker.device_ptr, blk), dev.wrap_buffer(blocker.device_ptr + blk, blk)
    times = {(form, mode): [] for form in ("three", "fused") for mode in ("idle", "queued")}
    for r in range(args.reps + 5):
        for mode in ("idle", "queued"):
            for form in ("three", "fused"):
                dev.fill_synthetic(bufs[form], n, scale=2.0 ** -20, encoder=enc)
                enc.synchronize()
                a, b = fw.Event(dev), fw.Event(dev)
                if mode == "queued":
                    dev.calib_copy(bdst, bsrc, blk, encoder=enc)
                a.record(enc)
                plans[("Forward", form)].proc(enc)
                b.record(enc)
                us = a.elapsed_ms(b) * 1e3
                if r >= 5:
                    times[(form, mode)].append(us)
    for form in ("three", "fused"):
        line = {"form": form, "launches": plans[("Forward", form)].get("launches_per_exec"), "reps": args.reps}
        for mode in ("idle", "queued"):
            t = sorted(times[(form, mode)])
            line[mode + "_us_median"] = round(t[len(t) // 2], 2)
            line[mode + "_us_min"] = round(t[0], 2)
            line[mode + "_us_p90"] = round(t[(len(t) * 9) // 10], 2)
        print(json.dumps(line), flush=True)
    print(json.dumps({"device_error_after_timing": plans[("Forward", "fused")].get("device_error")}), flush=True)


if __name__ == "__main__":
    main()
