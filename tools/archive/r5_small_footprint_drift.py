import json, os, sys
sys.path.insert(0, os.getcwd())
import fft_wgpu_amd as fw
dev, queue = fw.prepare_gpu(0)
enc = dev.create_command_encoder()
tot = 28
buf = dev.create_buffer(8 << tot)
for lg in (11, 12, 13, 14, 11, 13):
    n = 1 << lg; batch = 1 << (tot - lg)
    plan = fw.Forward(dev, queue, buf, n)
    times = []
    for r in range(40):
        dev.fill_synthetic(buf, n, scale=2.0 ** -20, encoder=enc)
        a, b = fw.Event(dev), fw.Event(dev)
        a.record(enc); plan.proc(enc); b.record(enc)
        times.append(round(a.elapsed_ms(b), 3))
    print(json.dumps({"lg_n": lg, "ms": times}), flush=True)
    plan.destroy()
