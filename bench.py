#!/usr/bin/env python3
"""bench.py -- headline benchmark: 1-D c2c fp32 forward FFT, N = 2^20, batch = 4096 per GPU.

One step = one `Forward.proc` over the whole per-GPU batch (synthetic interleaved complex fp32,
generated on the device, resident in HBM before the clock starts).  One process per GPU; batches
shard as independent slabs (no data-path collective; reference src/kernel/fft4.wgsl:21-23: one
`offset` per workgroup), so scaling is "weak": every rank runs the full 4096-transform slab
(config C4 = 4096 per GPU x 8).

Launch: under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) it uses
the environment it is given (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Without one, `python bench.py --gpus N`
starts its own N ranks -- fresh child processes, started before anything touches a GPU -- and supervises them:
the first child that exits non-zero takes its siblings down (SIGTERM, then SIGKILL) and the parent returns that
status, so a dead rank never leaves the others parked in a barrier.  It refuses to run with fewer visible
devices than --gpus.

Order of one run: rank 0 times the CPU baseline first (the oracle's restatement of the reference algorithm on
the host cores, ~12 s, nothing on the GPU yet; the other ranks wait in the rendezvous, timeout 300 s), then the
GPU phase -- warm-up, the K timed steps, the stream-ceiling calibration and a 100-exec spread leg on rank 0 (HIP
events only, not part of `value`), the batch-1 configurations of BASELINE.json (`"configs"`: N = 1024 = C1's shape, C2 = 2^20,
C5 = 2^24, `--config-execs` execs each with HIP events, one exec of each checked against the fp64 DFT; rank 0, not part of
`value`), and for N > 1 the slab-movement leg (below).  The line carries "gpu_phase_s".

Timing: K steps inside barrier + torch.cuda.synchronize() brackets, wall clock, MAX over ranks.
Because a forward FFT multiplies the RMS by 2^10 and `proc` works in place, the K steps run in
chunks of <= 8 with the input regenerated (at scale 2^-40) between chunks, outside the brackets,
so fp32 never overflows to inf/NaN (benchmarks on degenerate data are not representative).
HIP events on the launch stream time each step for the roofline object; for N > 1 the line carries every rank's
own time ("per_rank_ms": its wall clock between the opening barrier and its own synchronize, before the closing
barrier) and `roofline` is that of the SLOWEST rank.

Movement (N > 1, SURVEY.md 8(e): "reported as a separate line", never part of `value`): after the timed steps
rank 0 scatters one slab of MOVE_TRANSFORMS transforms to every rank and gathers them back through the library's
own communicator (`fwa_comm_scatter` / `fwa_comm_gather`: grouped RCCL send / receive over xGMI), timed with HIP
events; a watchdog prints the line without it if the exchange does not finish in MOVE_TIMEOUT_S.

Rehearsal (FWA_BENCH_REHEARSAL=1, never set by the driver): every rank runs on device 0 and the process group is
"gloo" (RCCL refuses two ranks on one device), so the whole N > 1 control flow -- rendezvous, barriers, MAX /
all-gather, rank-0-only printing, slab offsets -- executes on a one-GPU box.  The line says `"rehearsal": true`
in `config`, and its value is N ranks sharing ONE GPU: not a measurement of anything.
"""
import argparse
import json
import os
import signal
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 16      # one 8-B read + one 8-B write per complex sample
CHUNK = 8                       # steps between input regenerations (2^-40 * 2^(10*8) stays finite)
RENDEZVOUS_TIMEOUT_S = 300      # init_process_group / collectives: a missing rank is an error after this long.  Generous on
                                # purpose: on a cold box the ranks' first `import torch` takes 1-2 minutes and rank 0 spends
                                # 12 s more on the CPU baseline; a rank that DIES is caught at once by the launcher (torchrun,
                                # or supervise() below), not by this timeout
MOVE_TRANSFORMS = 256           # slab moved per rank by the movement leg (2 GiB at N = 2^20)
MOVE_TIMEOUT_S = 120


def usable_cores():
    """CPU share of this process: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def rehearsal():
    return os.environ.get("FWA_BENCH_REHEARSAL") == "1"


def supervise(procs, poll_s=0.1, grace_s=5.0, log=sys.stderr):
    """Wait for the ranks in `procs` (subprocess.Popen).  All exit 0 -> 0.  The first one that exits non-zero (or is
    killed by a signal) ends the run: its siblings get SIGTERM, after `grace_s` SIGKILL, and its status (signal n ->
    128 + n) is returned -- the survivors would otherwise sit in a barrier until the process-group timeout."""
    alive = list(procs)
    rc = 0
    while alive and rc == 0:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0:
                rc = r if r > 0 else 128 - r
                print(f"bench.py: rank {procs.index(p)} exited with status {r}; stopping the other {len(alive)} rank(s)",
                      file=log, flush=True)
                break
        if alive and rc == 0:
            time.sleep(poll_s)
    if alive:
        for p in alive:
            try:
                p.terminate()
            except OSError:
                pass
        deadline = time.monotonic() + grace_s
        for p in alive:
            try:
                p.wait(max(0.0, deadline - time.monotonic()))
            except Exception:
                try:
                    p.kill()
                except OSError:
                    pass
                p.wait()
    return rc


def self_launch(args):
    """No launcher environment and --gpus N > 1: start N ranks of this script (one process per GPU) and relay
    their exit status.  Runs before any HIP call; torch.cuda.device_count() does not initialise the GPU."""
    import socket
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    need = 1 if rehearsal() else args.gpus
    if visible < need:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {visible} device(s) visible; refusing to "
                         f"report a {args.gpus}-GPU figure from fewer GPUs")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

    def on_signal(signum, _frame):  # the parent is told to stop: the ranks must not outlive it
        for p in procs:
            if p.poll() is None:
                try:
                    p.terminate()
                except OSError:
                    pass
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    raise SystemExit(supervise(procs))


CONFIG_SHAPES = (("C1_shape", 10, "BASELINE.json configs[0] (its shape on the HIP path; the lavapipe leg cannot run offline)"),
                 ("C2", 20, "BASELINE.json configs[1]"),
                 ("C5", 24, "BASELINE.json configs[4]"))


def config_shapes(fw, dev, queue, enc, checker, reps):
    """The single-GPU batch-1 configurations of BASELINE.json beside the headline (rank 0, after the timed region, never part of
    `value`): N = 1024 (C1's shape), 2^20 (C2), 2^24 (C5), one transform each through `Forward.proc` (reference
    src/processor.rs:110-158).  Per shape: what the plan chose, HIP events around ONE `proc` on an idle stream (`reps` times;
    the input regenerated before every exec, outside the events), the same with a blocker queued first so that every launch of
    the exec is already in the queue when the device reaches the first event, Gsamples/s and the fraction of the 8 TB/s roofline
    from the idle-stream median, and a parity figure of ONE exec: max |y - r| / max |r| against the fp64 DFT of the same input
    (C1, C2; `checker` = the oracle's fwo_dft_f64 when the CPU-baseline leg loaded it, numpy.fft in float64 otherwise) or, at
    2^24, against the closed form of a unit impulse's transform exp(-2 pi i p k / n) evaluated in float64 (pins every output bin,
    the sign and the order without a 2^24-point CPU transform in the bench run)."""
    import numpy as np
    out = {}
    blk = 256 << 20
    blocker = dev.create_buffer(2 * blk)
    bsrc, bdst = dev.wrap_buffer(blocker.device_ptr, blk), dev.wrap_buffer(blocker.device_ptr + blk, blk)
    for name, lg, what in CONFIG_SHAPES:
        n = 1 << lg
        buf = dev.create_buffer(n * fw.COMPLEX_BYTES)
        plan = fw.Forward(dev, queue, buf, n)
        t = {}
        for mode in ("idle", "queued"):
            ts = []
            for r in range(reps + 5):
                dev.fill_synthetic(buf, n, scale=2.0 ** -20, encoder=enc)
                enc.synchronize()
                a, b = fw.Event(dev), fw.Event(dev)
                if mode == "queued":
                    dev.calib_copy(bdst, bsrc, blk, encoder=enc)
                a.record(enc)
                plan.proc(enc)
                b.record(enc)
                ms = a.elapsed_ms(b)
                if r >= 5:
                    ts.append(ms * 1e3)
            ts.sort()
            t[mode] = ts
        # parity of one exec
        if lg <= 20:
            dev.fill_synthetic(buf, n, encoder=enc)
            x = buf.map_read(stream=enc)
            y = plan.proc(enc).map_read(stream=enc)
            if checker is not None:
                ref, kind = checker.dft_f64(x, n, -1), "oracle.dft_f64 (fp64 DFT of the same input)"
            else:
                ref, kind = np.fft.fft(x.astype(np.complex128)), "numpy.fft float64 (--no-cpu-baseline: the oracle is not loaded)"
        else:
            p = 0x9E3779 % n | 1          # an odd position: every bin gets its own phase
            x = np.zeros(n, dtype=np.complex64)
            x[p] = 1.0
            queue.write_buffer(buf, 0, x, encoder=enc)
            y = plan.proc(enc).map_read(stream=enc)
            k = np.arange(n, dtype=np.int64)
            ang = (-2.0 * np.pi / n) * ((k * p) % n).astype(np.float64)
            ref, kind = np.cos(ang) + 1j * np.sin(ang), f"closed form of a unit impulse at {p}, float64"
        max_rel = float(np.abs(y.astype(np.complex128) - ref).max() / np.abs(ref).max())
        med = round(t["idle"][len(t["idle"]) // 2], 3)      # the figure the line prints; the rates below derive from it
        out[name] = {"config": what, "fft_len": n, "batch": 1, "plan_path": plan.get("path"), "factors": plan.get("factors"),
                     "launches": plan.get("launches_per_exec"), "execs": reps,
                     "us_median": med, "us_min": round(t["idle"][0], 3), "us_p90": round(t["idle"][(len(t["idle"]) * 9) // 10], 3),
                     "us_queued_median": round(t["queued"][len(t["queued"]) // 2], 3), "us_queued_min": round(t["queued"][0], 3),
                     "Gsamples_per_s": n / (med * 1e-6) / 1e9,
                     "frac": ALGO_BYTES_PER_SAMPLE * n / (med * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                     "parity_max_rel": max_rel, "parity_ok": bool(max_rel <= 1e-5), "parity_checker": kind}
        plan.destroy()
        buf.destroy()
    for b in (bsrc, bdst, blocker):
        b.destroy()
    out["_clock"] = ("HIP events around one Forward.proc on the launch stream; us_* = idle stream (the host's launch calls are inside the "
                     "interval: what a caller sees), us_queued_* = behind a 256-MiB blocker copy (device-side time of the launches alone); "
                     "frac = 16 B x N / us_median / 8 TB/s; batch-1 shapes live in the Infinity Cache and are launch-latency bound")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--fft-len", type=int, default=1 << 20)
    ap.add_argument("--batch", type=int, default=4096, help="transforms per GPU")
    ap.add_argument("--group", type=int, default=0, help="override plan tunable (0 = default)")
    ap.add_argument("--streams", type=int, default=0, help="override plan tunable (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spread", type=int, default=100, help="extra execs timed one by one after the K steps (0 = off)")
    ap.add_argument("--no-movement", action="store_true", help="N > 1: skip the slab scatter / gather leg")
    ap.add_argument("--config-execs", type=int, default=200,
                    help="execs per batch-1 configuration (C1's shape, C2, C5) timed after the headline on rank 0 (0 = off)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # never returns

    # the host driver of this pool only supports dmabuf IPC: RCCL's peer mappings fail with the legacy mode (exported on
    # the GPU boxes already; set here as well, before the runtime loads, so that a bare launcher environment works)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import torch
    import torch.distributed as dist

    world =int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("FWA_BENCH_FAIL_RANK") == str(rank):   # test knob: this rank dies before the rendezvous
        print(f"bench.py: rank {rank} exits on request (FWA_BENCH_FAIL_RANK)", file=sys.stderr, flush=True)
        sys.exit(3)
    rehearse = rehearsal() and world > 1
    device_index = 0 if rehearse else local_rank          # rehearsal: every rank shares device 0
    if torch.cuda.device_count() <= device_index:
        raise SystemExit(f"bench.py: rank {rank} has no device {device_index} ({torch.cuda.device_count()} visible)")

    # CPU baseline on rank 0 BEFORE the rendezvous and before anything runs on the GPU: afterwards the run is one
    # contiguous GPU phase.  The other ranks wait for rank 0's store inside init_process_group (RENDEZVOUS_TIMEOUT_S).
    cpu = None
    checker = None     # the oracle, when this run loaded it: also checks one exec of each batch-1 configuration (config_shapes)
    if rank == 0 and not args.no_cpu_baseline:
        import oracle  # checker only: times the CPU restatement of the reference algorithm beside the GPU number
        checker = oracle
        cores = usable_cores()
        cb = max(cores, 8)
        sps, reps = oracle.bench_forward(args.fft_len, cb, threads=cores, min_seconds=args.cpu_seconds)
        cpu = {"value": sps / 1e9, "unit": "Gsamples/s", "cores": cores, "kind": "port",
               "sample": f"{cb} transforms of N={args.fft_len} (same generator, seed 0x5EED), best of {reps} repetitions, "
                         f"OpenMP over transforms; CPU restatement of the reference radix-2 Stockham algorithm; timed "
                         f"on rank 0 before the GPU phase"}

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or os.environ.get("FWA_BENCH_FORCE_DIST") == "1"  # the env knob exercises the N>1 code path on one GPU
    backend = "gloo" if rehearse else "nccl"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        kw = {} if backend == "gloo" else {"device_id": torch.device("cuda", device_index)}
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S), **kw)
    coll_dev = torch.device("cpu") if backend == "gloo" else torch.device("cuda", device_index)

    def dist_barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[device_index])
        else:
            dist.barrier()

    def all_gather_f64(x):
        """every rank's scalar, in rank order"""
        if not use_dist:
            return [float(x)]
        out = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(out, torch.tensor([x], dtype=torch.float64, device=coll_dev))
        return [float(v.item()) for v in out]

    t_gpu_phase = time.perf_counter()
    import fft_wgpu_amd as fw
    from fft_wgpu_amd import sharding

    got = fw.prepare_gpu(device_index)
    if got is None:
        raise SystemExit("no usable gfx950 device")
    dev, queue = got
    n, batch = args.fft_len, args.batch
    nbytes = n * batch * fw.COMPLEX_BYTES
    buf = dev.create_buffer(nbytes)
    plan = fw.Forward(dev, queue, buf, n)
    if args.group:
        plan.set("group", args.group)
    if args.streams:
        plan.set("streams", args.streams)
    enc = dev.create_command_encoder()   # the stream every kernel of the timed region is launched on
    first, last = sharding.slab(batch * world, rank, world)   # this rank's slab of the global batch (config C4)
    assert last - first == batch

    def regen():
        dev.fill_synthetic(buf, n, first_transform=first, scale=2.0 ** -40, encoder=enc)
        enc.synchronize()

    def local_sync():
        enc.synchronize()
        torch.cuda.synchronize()

    def barrier():
        local_sync()
        if use_dist:
            dist_barrier()

    regen()
    for w in range(args.warmup):
        if w and w % CHUNK == 0:
            regen()            # as in the timed loop: never more than CHUNK in-place transforms on one fill (x 2^10 each)
        plan.proc(enc)
    barrier()

    ev = [(fw.Event(dev), fw.Event(dev)) for _ in range(args.steps)]
    wall = 0.0        # between the brackets, closing barrier included: MAX over ranks -> `value`
    wall_own = 0.0    # this rank's own work: opening barrier -> its own synchronize
    done = 0
    while done < args.steps:
        k = min(CHUNK, args.steps - done)
        regen()
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            ev[done + i][0].record(enc)
            plan.proc(enc)
            ev[done + i][1].record(enc)
        local_sync()
        t1 = time.perf_counter()
        if use_dist:
            dist_barrier()
        wall += time.perf_counter() - t0
        wall_own += t1 - t0
        done += k
    step_ms_events = [a.elapsed_ms(b) for a, b in ev]
    ev_ms = sum(step_ms_events) / len(step_ms_events)

    t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    per_rank_ms = all_gather_f64(wall_own / args.steps * 1e3)
    per_rank_ev_ms = all_gather_f64(ev_ms)
    per_rank_first = all_gather_f64(float(first))

    # calibration on the same stream, same run (rank 0): what a single streaming pass sustains on this box --
    #  * in place (every line read, then written back: the traffic of the one-launch FFT kernels; fwa_calib_copy with
    #    dst == src runs the normalize kernel: one workgroup per 64-KiB chunk, every wave walks 16 KiB, 32 nt loads in flight),
    #  * out of place (8 GiB -> another 8 GiB, the same launch shape).
    # With N > 1 in rehearsal the ranks share one GPU: calibration figures would mean nothing and are skipped.
    copy_gbps = stream_gbps = None
    spread = single_pass = shapes = None
    if rank == 0 and not rehearse:
        def rate(dst, src, nb, reps=3):
            dev.calib_copy(dst, src, nb, encoder=enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            for _ in range(reps):
                dev.calib_copy(dst, src, nb, encoder=enc)
            b.record(enc)
            return reps * 2 * nb / (a.elapsed_ms(b) * 1e-3) / 1e9
        half = min(nbytes // 2, 8 << 30) // 16 * 16
        if half >= (1 << 20):
            src_view = dev.wrap_buffer(buf.device_ptr, half)
            dst_view = dev.wrap_buffer(buf.device_ptr + half, half)
            copy_gbps = rate(dst_view, src_view, half)
            whole = dev.wrap_buffer(buf.device_ptr, 2 * half)
            stream_gbps = rate(whole, whole, 2 * half)
        # what a ONE-pass transform reaches on this box in this run: n = 1024 over the same 32 GiB (16 B/sample cross the
        # fabric once; the 2^20 transform needs two passes = 32 B/sample)
        if n > 1024 and nbytes % (1024 * 8) == 0:
            regen()
            small = fw.Forward(dev, queue, buf, 1024)
            small.proc(enc)
            a, b = fw.Event(dev), fw.Event(dev)
            a.record(enc)
            for _ in range(2):
                small.proc(enc)
            b.record(enc)
            ms1 = a.elapsed_ms(b) / 2
            single_pass = {"fft_len": 1024, "batch": nbytes // 8 // 1024, "ms": ms1,
                           "GBps": ALGO_BYTES_PER_SAMPLE * (nbytes // 8) / (ms1 * 1e-3) / 1e9,
                           "frac": ALGO_BYTES_PER_SAMPLE * (nbytes // 8) / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            small.destroy()
        # spread leg: `args.spread` more execs timed one by one with HIP events (not part of `value`)
        if args.spread > 0:
            sp = []
            done_s = 0
            while done_s < args.spread:
                k = min(CHUNK, args.spread - done_s)
                regen()
                evs = [(fw.Event(dev), fw.Event(dev)) for _ in range(k)]
                for a, b in evs:
                    a.record(enc)
                    plan.proc(enc)
                    b.record(enc)
                enc.synchronize()
                sp += [a.elapsed_ms(b) for a, b in evs]
                done_s += k
            sp.sort()
            spread = {"execs": len(sp), "ms_p10": sp[len(sp) // 10], "ms_p50": sp[len(sp) // 2], "ms_p90": sp[(len(sp) * 9) // 10],
                      "ms_min": sp[0], "ms_max": sp[-1]}
        # the other single-GPU configurations of BASELINE.json (batch 1: latency shapes), after everything the headline needs
        if args.config_execs > 0 and n == (1 << 20):
            shapes = config_shapes(fw, dev, queue, enc, checker, args.config_execs)
    enc.synchronize()
    per_rank_ev_min = all_gather_f64(min(step_ms_events))

    def build_line():
        samples_per_step = n * batch * world
        ms_per_step = wall_max / args.steps * 1e3
        value = samples_per_step / (wall_max / args.steps) / 1e9
        slow = max(range(world), key=lambda r: per_rank_ev_ms[r]) if use_dist else 0
        slow_ms = per_rank_ev_ms[slow] if use_dist else ev_ms       # roofline of the slowest rank's GPU
        achieved = ALGO_BYTES_PER_SAMPLE * n * batch / (slow_ms * 1e-3) / 1e9
        # Figures that cannot be measured inside the timed run (PMC counters need their own rocprofv3 passes; the
        # linear-stream floors are probe programs): read from profiles/bench_reference.json, each with its source.
        traffic = traffic_source = floors = isolated = taken_on = None
        traffic_stale = None
        tpath = os.path.join(ROOT, "profiles", "bench_reference.json")
        if os.path.exists(tpath):
            try:
                ref = json.load(open(tpath))
                traffic = ref.get("traffic_bytes_per_exec", {}).get(f"{n}x{batch}")
                traffic_source = ref.get("traffic_source") if traffic else None
                # the stored figure belongs to the kernel sources it was profiled on: say so when they have changed since
                taken_on = ref.get("taken_on") if traffic else None
                if traffic:
                    import hashlib
                    h = hashlib.sha256()
                    for rel in (taken_on or {}).get("kernel_sources", ["fft_wgpu_amd/csrc/tile_1m.h", "fft_wgpu_amd/csrc/kernels_1m.hip"]):
                        h.update(open(os.path.join(ROOT, rel), "rb").read())
                    traffic_stale = (taken_on or {}).get("kernel_source_sha256") != h.hexdigest()
                if n == (1 << 20):
                    floors = ref.get("measured_floors")
                    isolated = ref.get("isolated_kernel_us")
            except Exception:
                traffic = None
        launches = max(1, plan.get("launches_per_exec"))
        chains = max(1, plan.get("streams"))
        group = plan.get("group")
        # one exec = `launches` kernel launches (k_p1_1m then k_p2_1m per group of transforms) on `chains` concurrent
        # internal streams; a launch carries one pass over `group` transforms = half of their algorithmic bytes.
        avg_launch_us = slow_ms * 1e3 * chains / launches
        launch_bytes = ALGO_BYTES_PER_SAMPLE * n * group // 2
        return {
            "metric": "Gsamples/s, 1-D c2c fp32 forward FFT N=2^20 batch=4096 per GPU",
            "value": value, "unit": "Gsamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"1-D c2c fp32 FFT N={n} batch={batch} per GPU (BASELINE.json configs[2]"
                                   f"{'; configs[3] shape' if world > 1 else ''})",
                       "fft_len": n, "batch_per_gpu": batch, "parallelism": f"batch-sharded x{world}, no collective",
                       "rehearsal": bool(rehearse),
                       "dist_backend": (dist.get_backend() if use_dist else None),
                       "dist_world_size": (dist.get_world_size() if use_dist else None),
                       "slab_first_transform_per_rank": [int(v) for v in per_rank_first],
                       "chain_streams_checked": dev.stats().get("chain_checks"), "chain_streams_rejected": dev.stats().get("chain_rejects"),
                       "plan_path": plan.get("path"), "group": group, "streams": chains,
                       "tile_w": plan.get("tile_w"), "launches_per_step": launches,
                       "scratch_bytes": plan.get("scratch_bytes")},
            # every rank's own step time (wall clock, opening barrier -> its own synchronize) and HIP-event exec time
            "per_rank_ms": per_rank_ms, "per_rank_ms_min": min(per_rank_ms), "per_rank_ms_max": max(per_rank_ms),
            "per_rank_hip_event_ms": per_rank_ev_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "rank": slow, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_stale": traffic_stale, "traffic_taken_on": taken_on,
                         "measured_floors": floors, "isolated_kernel_us": isolated,
                         "kernel": "k_p1_1m + k_p2_1m (the launches of one fwa_plan_exec: pass 1 then pass 2 per group "
                                   "of transforms, alternating over the chains)",
                         "ms_per_exec_hip_events": slow_ms, "ms_min": per_rank_ev_min[slow],
                         # rocprofv3's mean duration over the k_p1_1m and k_p2_1m launches must equal avg_launch_us
                         # (launches of different chains overlap: sum of durations / chains = exec time)
                         "per_launch": {"launches": launches, "chains": chains, "avg_launch_us": avg_launch_us,
                                        "algorithmic_bytes": launch_bytes,
                                        "achieved_GBps_one_launch": launch_bytes / (avg_launch_us * 1e-6) / 1e9,
                                        "traffic_bytes": (traffic / launches) if traffic else None},
                         "algorithmic_bytes_per_exec": ALGO_BYTES_PER_SAMPLE * n * batch,
                         # what ONE streaming pass over the data sustains in this run: in place (the one-launch FFT kernels'
                         # traffic) and out of place; a two-pass transform moves 32 B/sample at about this rate
                         "copy_ceiling_GBps_same_run": stream_gbps, "copy_ceiling_kind": "in-place read + write-back of 16 GiB; one workgroup per 64-KiB chunk, every wave walks 16 KiB, 32 nt 8-byte loads in flight per thread (profiles/round4/probe_stream_shapes.txt)",
                         "copy_out_of_place_GBps_same_run": copy_gbps,
                         "single_pass_reference_same_run": single_pass,
                         "exec_spread_hip_events": spread},
            # C1's shape, C2 and C5 (batch 1) under the same clock, same run; never part of `value`
            "configs": shapes,
            "cpu_baseline": cpu,
            "movement": None,
            "gpu_phase_s": time.perf_counter() - t_gpu_phase,
        }

    # The line is complete BEFORE the movement leg starts: if that leg hangs, the watchdog thread only fills in `movement` and
    # prints -- it makes no library or HIP call while the main thread is parked in a sync on a possibly wedged device.
    line = build_line() if rank == 0 else None

    # ---- movement leg (N > 1; never part of `value`) -------------------------------------------------------------------
    movement = None
    if use_dist and not args.no_movement:
        if rehearse:
            movement = {"skipped": "rehearsal: RCCL refuses two ranks on one device (fwa_comm_* needs one GPU per rank)"}
        else:
            mt = min(MOVE_TRANSFORMS, batch // (world + 1))     # full batch (world slabs) + this rank's slab fit `buf`
            if mt < 1:
                movement = {"skipped": f"batch {batch} per GPU is too small to carve {world} + 1 slabs out of the buffer"}
            else:
                finished = threading.Event()

                def leave(what):
                    """The exchange failed or hangs on this rank: the other ranks may be parked in it for good, so no
                    further collective is safe.  The timed figures are complete: rank 0 prints the line with the failure
                    in `movement`, every rank leaves with status 0 (the measurement stands; the failure is in the line)."""
                    if rank == 0:
                        line["movement"] = {"error": what}
                        print(json.dumps(line), flush=True)
                    else:
                        time.sleep(3.0)   # rank 0 prints first; a launcher may tear the job down at the first exit
                    os._exit(0)

                def give_up():
                    if not finished.is_set():
                        leave(f"fwa_comm scatter / gather did not finish in {MOVE_TIMEOUT_S} s; the line is complete without it")
                dog = threading.Timer(MOVE_TIMEOUT_S, give_up)
                dog.daemon = True
                dog.start()
                try:
                    movement = move_slabs(fw, dist, torch, dev, enc, buf, n, mt, rank, world, coll_dev)
                except Exception as e:   # a status code from the library on THIS rank
                    finished.set()
                    dog.cancel()
                    leave(f"{type(e).__name__}: {e}")
                finished.set()
                dog.cancel()

    if rank == 0:
        line["movement"] = movement
        line["gpu_phase_s"] = time.perf_counter() - t_gpu_phase
        print(json.dumps(line), flush=True)
    if use_dist:
        # the line is out: a closing barrier / communicator teardown that hangs must not turn a finished measurement into a
        # run killed at its time limit
        bye = threading.Timer(60.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        dist_barrier()
        dist.destroy_process_group()
        bye.cancel()


def move_slabs(fw, dist, torch, dev, enc, buf, n, mt, rank, world, coll_dev):
    """Rank 0 scatters `mt` transforms to every rank and gathers them back through fwa_comm_* (RCCL send / receive,
    grouped), HIP events on the launch stream of each rank; the rates are bytes that LEAVE / ENTER rank 0 over xGMI
    (its own slab is a local copy) divided by the slowest rank's time.  Buffers are views of the benchmark buffer:
    on rank 0 the first world * mt transforms are the full batch and the next mt its slab; elsewhere the first mt."""
    from fft_wgpu_amd import sharding
    if os.environ.get("FWA_BENCH_FAIL_MOVEMENT") == str(rank):    # test knob: the library refuses on this rank
        raise RuntimeError("movement failure on request (FWA_BENCH_FAIL_MOVEMENT)")
    tb = 8 * n
    uid = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
    if rank == 0:
        uid = torch.frombuffer(bytearray(sharding.Comm.unique_id()), dtype=torch.uint8).to(coll_dev)
    dist.broadcast(uid, 0)
    comm = sharding.Comm(dev, bytes(uid.cpu().numpy().tobytes()), world, rank)
    full = dev.wrap_buffer(buf.device_ptr, world * mt * tb) if rank == 0 else None
    slab = dev.wrap_buffer(buf.device_ptr + (world * mt * tb if rank == 0 else 0), mt * tb)
    gbatch = world * mt
    out = {"transforms_per_rank": mt, "bytes_per_rank": mt * tb, "root": 0, "api": "fwa_comm_scatter / fwa_comm_gather"}
    for name, call in (("scatter", lambda: comm.scatter(full, slab, n, gbatch, root=0, encoder=enc)),
                       ("gather", lambda: comm.gather(slab, full, n, gbatch, root=0, encoder=enc))):
        call()                      # first exchange: connection set-up of the communicator
        enc.synchronize()
        a, b = fw.Event(dev), fw.Event(dev)
        a.record(enc)
        call()
        b.record(enc)
        ms = torch.tensor([a.elapsed_ms(b)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        ms = float(ms.item())
        out[name + "_ms"] = ms
        out[name + "_GBps"] = (world - 1) * mt * tb / (ms * 1e-3) / 1e9 if world > 1 else None
    comm.destroy()
    return out


if __name__ == "__main__":
    main()
