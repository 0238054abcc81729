#!/bin/bash
# round 4 experiment: pass 1 of group g+1 and pass 2 of group g fused into one launch (key "fused": 1 plain, 2 with the pair map)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4fused; mkdir -p $O
timeout -k 10 300 python3 - <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import fft_wgpu_amd as fw, oracle
dev, queue = fw.prepare_gpu(0)
enc = dev.create_command_encoder()
n = 1 << 20
for batch in (96, 100, 33, 16):
    x = oracle.gen_input(n, batch, first_transform=3)
    def run(**kw):
        src = dev.create_buffer(x.nbytes); queue.write_buffer(src, 0, x)
        plan = fw.Forward(dev, queue, src, n)
        for k, v in kw.items(): plan.set(k, v)
        y = plan.proc(enc).map_read(stream=enc)
        return y
    ref = run()
    mx, _ = oracle.compare(ref[:n], oracle.dft_f64(x[:n], n, -1)); assert mx <= 1e-5
    for f in (1, 2):
        for rep in range(2):
            y = run(fused=f, streams=2)
            bad = np.flatnonzero(y.view(np.uint64) != ref.view(np.uint64))
            assert bad.size == 0, (batch, f, rep, bad.size, bad[:4])
print("fused launches bit-identical")
PY
rm -f $O/sweep.jsonl
for rep in 1 2; do
  timeout -k 10 300 python3 tools/sweep.py --lg 20 --batch 4096 --reps 7 --set "" --set "fused=1" --set "fused=2" --set "streams=1" >> $O/sweep.jsonl
done
timeout -k 10 300 python3 tools/sweep.py --lg 20 --batch 256 --reps 9 --set "" --set "fused=1" --set "fused=2" >> $O/sweep.jsonl
python3 - <<PY
import json
for l in open("$O/sweep.jsonl"):
    d = json.loads(l); print(d["batch"], "%-12s" % d["setting"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
