import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_wgpu_amd as fw
dev, queue = fw.prepare_gpu(0)
enc = dev.create_command_encoder()
for lg, batch in ((16, 4096), (24, 16), (21, 128)):
    n = 1 << lg
    buf = dev.create_buffer(n * batch * 8)
    for dbg in (0, 1):
        for streams in (1, 2, 3):
            plan = fw.Forward(dev, queue, buf, n)
            plan.set("dbg", dbg); plan.set("streams", streams)
            ts = []
            for r in range(4):
                dev.fill_synthetic(buf, n, scale=2.0 ** -20, encoder=enc)
                a, b = fw.Event(dev), fw.Event(dev)
                a.record(enc); plan.proc(enc); b.record(enc)
                ts.append(a.elapsed_ms(b))
            print(json.dumps({"lg": lg, "skip_twiddle": dbg, "streams": streams, "ms": sorted(ts)[1]}), flush=True)
            plan.destroy()
    buf.destroy()
