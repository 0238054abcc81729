#!/usr/bin/env python3
"""tools/fused_c2_stamps.py -- where the one-launch form of config C2 spends its time: six s_memrealtime stamps (100 MHz) per
workgroup (entry, end of pass A, out of barrier 1, end of pass B, out of barrier 2, end of pass C), key "fused" = 2."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402

dev, queue = fw.prepare_gpu(0, lab=True)
enc = dev.create_command_encoder()
n = 1 << 20
buf = dev.create_buffer(n * 8)
p = fw.Forward(dev, queue, buf, n)
p.set("fused", 2)
ctl = dev.wrap_buffer(p.get("ctl_ptr"), (64 + 256) * 4 + 256 * 64)
for rep in range(8):
    dev.fill_synthetic(buf, n, scale=2.0 ** -20, encoder=enc)
    enc.synchronize()
    p.proc(enc)
    enc.synchronize()
    raw = ctl.map_read(stream=enc, dtype=np.uint64)
    st = raw[(64 + 256) * 4 // 8:][:256 * 8].reshape(256, 8)[:, :6].astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    seg = {"entry_spread": float(us[:, 0].max()), "passA_end_min": float(us[:, 1].min()), "passA_end_max": float(us[:, 1].max()),
           "barrier1_out_min": float(us[:, 2].min()), "barrier1_out_max": float(us[:, 2].max()),
           "passB_end_min": float(us[:, 3].min()), "passB_end_max": float(us[:, 3].max()),
           "barrier2_out_min": float(us[:, 4].min()), "barrier2_out_max": float(us[:, 4].max()),
           "passC_end_min": float(us[:, 5].min()), "passC_end_max": float(us[:, 5].max()),
           "own_passA_median": float(np.median(us[:, 1] - us[:, 0])), "own_passB_median": float(np.median(us[:, 3] - us[:, 2])),
           "own_passC_median": float(np.median(us[:, 5] - us[:, 4]))}
    print(json.dumps({"rep": rep, **{k: round(v, 2) for k, v in seg.items()}}), flush=True)
print(json.dumps({"device_error": p.get("device_error")}))
