import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fft_wgpu_amd as fw
import oracle
dev, queue = fw.prepare_gpu(0)
for lg, batch, mt in ((16,1,None),(16,2,None),(16,8,None),(16,9,None),(16,16,None),(16,100,None),(16,100,1),(16,100,3),(17,5,None),(18,3,None)):
    n = 1 << lg
    x = oracle.gen_input(n, batch, first_transform=lg)
    src = dev.create_buffer(x.nbytes); queue.write_buffer(src, 0, x)
    plan = fw.Forward(dev, queue, src, n)
    plan.set("path", 8)
    if mt: plan.set("max_teams", mt)
    enc = dev.create_command_encoder()
    t0 = time.perf_counter(); out = plan.proc(enc); enc.synchronize(); dt = time.perf_counter() - t0
    y = out.map_read(stream=enc)
    r = oracle.dft_f64(x, n, -1)
    worst = max(oracle.compare(y[t*n:(t+1)*n], r[t*n:(t+1)*n])[0] for t in range(batch))
    print(lg, batch, "max_teams", plan.get("max_teams"), "wgs", plan.get("wgs"), "err", plan.get("device_error"), "ms %.2f" % (dt*1e3), "worst %.2e" % worst, flush=True)
