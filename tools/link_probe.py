#!/usr/bin/env python3
"""tools/link_probe.py -- host-link rates through the C ABI: pinned H2D alone, D2H alone, both at once (two streams),
for several transfer sizes; run it again with HSA_ENABLE_SDMA=0 to compare the SDMA engines with shader copies."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def main():
    dev, queue = fw.prepare_gpu(0)
    up, down = dev.create_command_encoder(), dev.create_command_encoder()
    for mb in (1, 10, 64, 256):
        count = mb * (1 << 20) // 8
        hin, hout = dev.pinned_array(count), dev.pinned_array(count)
        hin[:] = 1
        a, b = dev.create_buffer(count * 8), dev.create_buffer(count * 8)
        reps = max(4, 2048 // mb)
        res = {"MiB": mb, "sdma": os.environ.get("HSA_ENABLE_SDMA", "default")}
        for name, do_up, do_down in (("h2d", 1, 0), ("d2h", 0, 1), ("both", 1, 1)):
            for r in range(reps + 2):
                if r == 2:
                    up.synchronize(); down.synchronize()
                    t0 = time.perf_counter()
                if do_up:
                    queue.write_buffer(a, 0, hin, encoder=up)
                if do_down:
                    dev.download_async(hout, b, down)
            up.synchronize(); down.synchronize()
            dt = time.perf_counter() - t0
            res[name + "_GBps_each_way"] = round(count * 8 * reps / dt / 1e9, 2)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
