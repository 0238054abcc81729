"""tools/variant_parity_check.py -- parity of a VARIANT library of the 2^20 pipeline (an experiment built with extra -D flags from a
patch under profiles/round4/*.patch and loaded through FWA_LAB_LIBRARY): forward against the f64 DFT, round trip, two geometries."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fft_wgpu_amd as fw, oracle
dev, queue = fw.prepare_gpu(0, lab=True)
enc = dev.create_command_encoder()
n = 1 << 20
for batch, kw in ((5, {}), (21, {"group": 4, "streams": 2})):
    x = oracle.gen_input(n, batch, first_transform=7)
    src = dev.create_buffer(x.nbytes); queue.write_buffer(src, 0, x)
    plan = fw.Forward(dev, queue, src, n)
    for k, v in kw.items(): plan.set(k, v)
    assert plan.get("path") == 1
    y = plan.proc(enc).map_read(stream=enc)
    r = oracle.dft_f64(x[:2 * n], n, -1)
    for t in range(2):
        mx, l2 = oracle.compare(y[t*n:(t+1)*n], r[t*n:(t+1)*n]); assert mx <= 1e-5, mx
    z = fw.Inverse(dev, queue, src, n).proc(enc).map_read(stream=enc)
    assert np.abs(z - x).max() <= 1e-5 * np.abs(x).max()
print("walk variant parity ok", mx)
