"""tools/pipe_probe.py -- which stream/event structure lets the reference loop (upload -> proc -> copy -> read-back) overlap its
stages on this ROCm stack?  One JSON line per structure."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fft_wgpu_amd as fw
dev, queue = fw.prepare_gpu(0)
n, batch = 512, 2500
count = n*batch; nbytes = count*8
A, B = dev.create_command_encoder(), dev.create_command_encoder()
hin = [dev.pinned_array(count) for _ in range(2)]; hout = [dev.pinned_array(count) for _ in range(2)]
for h in hin: h[:] = 1
src = [dev.create_buffer(nbytes) for _ in range(2)]; stg = [dev.create_buffer(nbytes) for _ in range(2)]
plans = [fw.Forward(dev, queue, b, n) for b in src]
ev = [fw.Event(dev) for _ in range(2)]
def run(name, body, iters=300):
    for it in range(iters+4):
        if it == 4:
            A.synchronize(); B.synchronize(); t0 = time.perf_counter()
        body(it)
    A.synchronize(); B.synchronize()
    dt = time.perf_counter()-t0
    print(json.dumps({"what": name, "it_per_s": round(iters/dt,1), "ms_per_it": round(dt/iters*1e3,4)}), flush=True)
def up_only(it):
    s = it&1; queue.write_buffer(src[s], 0, hin[s], encoder=A)
def up_fft(it):
    s = it&1; queue.write_buffer(src[s], 0, hin[s], encoder=A); out = plans[s].proc(A); A.copy_buffer_to_buffer(out, 0, stg[s], 0, nbytes)
def down_only(it):
    s = it&1; dev.download_async(hout[s], stg[s], B)
def indep(it):
    up_fft(it); down_only(it)
def dep_one_way(it):
    s = it&1; up_fft(it); ev[s].record(A); B.wait_event(ev[s]); dev.download_async(hout[s], stg[s], B)
def fft_only(it):
    s = it&1; out = plans[s].proc(A); A.copy_buffer_to_buffer(out, 0, stg[s], 0, nbytes)
run("H2D only (SDMA)", up_only)
run("FFT + D2D only", fft_only)
run("H2D + FFT + D2D on one stream", up_fft)
run("D2H only (SDMA)", down_only)
run("independent: A = H2D+FFT+D2D, B = D2H", indep)
run("A -> B event dependency per iteration", dep_one_way)
ed = [fw.Event(dev) for _ in range(2)]
def dep_both(it):
    s = it&1
    queue.write_buffer(src[s], 0, hin[s], encoder=A)
    if it >= 2: A.wait_event(ed[s])
    out = plans[s].proc(A); A.copy_buffer_to_buffer(out, 0, stg[s], 0, nbytes)
    ev[s].record(A); B.wait_event(ev[s]); dev.download_async(hout[s], stg[s], B); ed[s].record(B)
run("A -> B and B -> A (slot reuse) event dependencies", dep_both)
eu = [fw.Event(dev) for _ in range(2)]
def dep_both_hostsync(it):
    s = it&1
    if it >= 2: eu[s].synchronize()
    queue.write_buffer(src[s], 0, hin[s], encoder=A); eu[s].record(A)
    if it >= 2: A.wait_event(ed[s])
    out = plans[s].proc(A); A.copy_buffer_to_buffer(out, 0, stg[s], 0, nbytes)
    ev[s].record(A); B.wait_event(ev[s]); dev.download_async(hout[s], stg[s], B); ed[s].record(B)
run("the same + host waits for the slot's previous upload", dep_both_hostsync)
def hostsync_reuse(it):
    s = it&1
    if it >= 2: ed[s].synchronize()      # host has the result of iteration it-2 (and staging[s] is free again)
    queue.write_buffer(src[s], 0, hin[s], encoder=A)
    out = plans[s].proc(A); A.copy_buffer_to_buffer(out, 0, stg[s], 0, nbytes)
    ev[s].record(A); B.wait_event(ev[s]); dev.download_async(hout[s], stg[s], B); ed[s].record(B)
run("A -> B event; slot reuse guarded by a HOST wait on the slot's read-back", hostsync_reuse)
ed3 = [fw.Event(dev) for _ in range(3)]; ev3 = [fw.Event(dev) for _ in range(3)]
hin3 = hin + [dev.pinned_array(count)]; hout3 = hout + [dev.pinned_array(count)]; hin3[2][:] = 1
src3 = src + [dev.create_buffer(nbytes)]; stg3 = stg + [dev.create_buffer(nbytes)]; plans3 = plans + [fw.Forward(dev, queue, src3[2], n)]
def hostsync_reuse3(it):
    s = it % 3
    if it >= 3: ed3[s].synchronize()
    queue.write_buffer(src3[s], 0, hin3[s], encoder=A)
    out = plans3[s].proc(A); A.copy_buffer_to_buffer(out, 0, stg3[s], 0, nbytes)
    ev3[s].record(A); B.wait_event(ev3[s]); dev.download_async(hout3[s], stg3[s], B); ed3[s].record(B)
run("the same with 3 slots", hostsync_reuse3)
pipe = fw.HostPipeline(dev, queue, lambda d, q, b: fw.Forward(d, q, b, n), count, slots=2)
for h in pipe.hin: h[:] = 1
def hp(it): pipe.submit()
run("fft_wgpu_amd.HostPipeline (2 slots)", hp)
