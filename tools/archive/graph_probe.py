#!/usr/bin/env python3
"""tools/graph_probe.py -- direct fwa_plan_exec against a replayed hipGraph of the same exec (captured on a wrapped
caller stream): does graph replay shorten the launch-bound shapes?  One JSON line per shape."""
import ctypes
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw  # noqa: E402


def hip_runtime():
    """The HIP runtime image already mapped into this process (the one the library uses)."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return ctypes.CDLL(line.split()[-1])
    raise SystemExit("no libamdhip64 mapped")


def main():
    dev, queue = fw.prepare_gpu(0)
    hip = hip_runtime()
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0
    enc = dev.create_command_encoder(hip_stream=stream)
    for lg, batch, reps in ((10, 1, 200), (16, 1, 200), (18, 1, 200), (20, 1, 200), (21, 1, 200), (24, 1, 100), (18, 64, 100),
                            (20, 64, 50), (20, 4096, 5)):
        n = 1 << lg
        buf = dev.create_buffer(n * batch * 8)
        plan = fw.Forward(dev, queue, buf, n)
        dev.fill_synthetic(buf, n, scale=2.0 ** -60, encoder=enc)
        plan.proc(enc)
        enc.synchronize()
        graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipStreamBeginCapture(stream, 0) == 0
        plan.proc(enc)
        assert hip.hipStreamEndCapture(stream, ctypes.byref(graph)) == 0
        assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
        res = {}
        for mode in ("direct", "graph", "direct", "graph"):
            times = []
            for r in range(reps + 3):
                if r % 2 == 0:
                    dev.fill_synthetic(buf, n, scale=2.0 ** -60, encoder=enc)
                a, b = fw.Event(dev), fw.Event(dev)
                a.record(enc)
                if mode == "direct":
                    plan.proc(enc)
                else:
                    assert hip.hipGraphLaunch(gexec, stream) == 0
                b.record(enc)
                if r >= 3:
                    times.append(a.elapsed_ms(b))
            res.setdefault(mode, []).append(sorted(times)[len(times) // 2] * 1e3)
        print(json.dumps({"lg_n": lg, "batch": batch, "launches": plan.get("launches_per_exec"),
                          "direct_us": [round(v, 2) for v in res["direct"]], "graph_us": [round(v, 2) for v in res["graph"]]}), flush=True)
        hip.hipGraphExecDestroy(gexec)
        hip.hipGraphDestroy(graph)
        plan.destroy()
        del buf


if __name__ == "__main__":
    main()
