import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import fft_wgpu_amd as fw
import oracle
dev, queue = fw.prepare_gpu(0)
for lg, batch in ((3,1000),(3,256),(3,300),(4,600),(5,300),(2,3000),(1,9000)):
    n = 1 << lg
    x = oracle.gen_input(n, batch, first_transform=lg)
    src = dev.create_buffer(x.nbytes); queue.write_buffer(src, 0, x)
    plan = fw.Forward(dev, queue, src, n)
    enc = dev.create_command_encoder()
    out = plan.proc(enc); y = out.map_read(stream=enc)
    r = oracle.dft_f64(x, n, -1)
    bad = [t for t in range(batch) if oracle.compare(y[t*n:(t+1)*n], r[t*n:(t+1)*n])[0] > 1e-5]
    print(lg, batch, "bad transforms:", len(bad), bad[:10], bad[-3:] if bad else "")
