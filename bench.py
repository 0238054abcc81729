#!/usr/bin/env python3
"""bench.py -- headline benchmark: 1-D c2c fp32 forward FFT, N = 2^20, batch = 4096 per GPU.

One step = one `Forward.proc` over the whole per-GPU batch (synthetic interleaved complex fp32,
generated on the device, resident in HBM before the clock starts).  One process per GPU; batches
shard as independent slabs (no data-path collective), so scaling is "weak": every rank runs the
full 4096-transform slab (config C4 = 4096 per GPU x 8).

Launch: `python bench.py --gpus N` starts its own N ranks (child processes, started before anything touches a
GPU) when WORLD_SIZE is not set; under torchrun it uses the environment it is given.  It refuses to run with
fewer visible devices than --gpus.

Order of one run (rank 0, N = 1): the CPU baseline first (the oracle's restatement of the reference algorithm on the host
cores, ~12 s, nothing on the GPU yet), then the GPU phase -- warm-up, the K timed steps, the stream-ceiling calibration
and a 100-exec spread leg (HIP events only, not part of `value`); the line carries "gpu_phase_s", the wall time of
that phase, so that a utilisation sampler's window can be compared with it.

Timing: K steps inside barrier + torch.cuda.synchronize() brackets, wall clock, MAX over ranks.
Because a forward FFT multiplies the RMS by 2^10 and `proc` works in place, the K steps run in
chunks of <= 8 with the input regenerated (at scale 2^-40) between chunks, outside the brackets,
so fp32 never overflows to inf/NaN (benchmarks on degenerate data are not representative).
HIP events on the launch stream time each step for the roofline object.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 16      # one 8-B read + one 8-B write per complex sample
CHUNK = 8                       # steps between input regenerations (2^-40 * 2^(10*8) stays finite)


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


def self_launch(args):
    """No launcher environment and --gpus N > 1: start N ranks of this script (one process per GPU) and relay
    their exit status.  Runs before any HIP call; torch.cuda.device_count() does not initialise the GPU."""
    import socket
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    if visible < args.gpus:
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
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


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
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # never returns

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("FWA_BENCH_FORCE_DIST") == "1"  # the env knob exercises the N>1 code path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import fft_wgpu_amd as fw
    from fft_wgpu_amd import sharding

    # CPU baseline BEFORE anything runs on the GPU (rank 0, N = 1 only): afterwards the run is one contiguous GPU phase
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        import oracle  # checker only: times the CPU restatement of the reference algorithm beside the GPU number
        cores = usable_cores()
        cb = max(cores, 8)
        sps, reps = oracle.bench_forward(args.fft_len, cb, threads=cores, min_seconds=args.cpu_seconds)
        cpu = {"value": sps / 1e9, "unit": "Gsamples/s", "cores": cores, "kind": "port",
               "sample": f"{cb} transforms of N={args.fft_len} (same generator, seed 0x5EED), best of {reps} repetitions, "
                         f"OpenMP over transforms; CPU restatement of the reference radix-2 Stockham algorithm; timed "
                         f"before the GPU phase"}
    t_gpu_phase = time.perf_counter()

    got = fw.prepare_gpu(local_rank)
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

    def barrier():
        enc.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    regen()
    for _ in range(args.warmup):
        plan.proc(enc)
    barrier()

    ev = [(fw.Event(dev), fw.Event(dev)) for _ in range(args.steps)]
    wall = 0.0
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
        barrier()
        wall += time.perf_counter() - t0
        done += k
    step_ms_events = [a.elapsed_ms(b) for a, b in ev]

    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    # calibration on the same stream, same run (rank 0): what a single streaming pass sustains on this box --
    #  * in place (every line read, then written back: the traffic of the one-launch FFT kernels; fwa_calib_copy with
    #    dst == src runs the normalize kernel: one workgroup per 64-KiB chunk, every wave walks 16 KiB, 32 nt loads in flight),
    #  * out of place (8 GiB -> another 8 GiB, the same launch shape).
    copy_gbps = stream_gbps = None
    spread = single_pass = None
    if rank == 0:
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
    enc.synchronize()
    gpu_phase_s = time.perf_counter() - t_gpu_phase

    if rank == 0:
        samples_per_step = n * batch * world
        ms_per_step = wall_max / args.steps * 1e3
        value = samples_per_step / (wall_max / args.steps) / 1e9
        ev_ms = sum(step_ms_events) / len(step_ms_events)
        achieved = ALGO_BYTES_PER_SAMPLE * n * batch / (ev_ms * 1e-3) / 1e9
        # Figures that cannot be measured inside the timed run (PMC counters need their own rocprofv3 passes; the
        # linear-stream floors are probe programs): read from profiles/bench_reference.json, each with its source.
        traffic = traffic_source = floors = isolated = None
        tpath = os.path.join(ROOT, "profiles", "bench_reference.json")
        if os.path.exists(tpath):
            try:
                ref = json.load(open(tpath))
                traffic = ref.get("traffic_bytes_per_exec", {}).get(f"{n}x{batch}")
                traffic_source = ref.get("traffic_source") if traffic else None
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
        avg_launch_us = ev_ms * 1e3 * chains / launches
        launch_bytes = ALGO_BYTES_PER_SAMPLE * n * group // 2
        line = {
            "metric": "Gsamples/s, 1-D c2c fp32 forward FFT N=2^20 batch=4096 per GPU",
            "value": value, "unit": "Gsamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"1-D c2c fp32 FFT N={n} batch={batch} per GPU (BASELINE.json configs[2]"
                                   f"{'; configs[3] shape' if world > 1 else ''})",
                       "fft_len": n, "batch_per_gpu": batch, "parallelism": f"batch-sharded x{world}, no collective",
                       "dist_backend": (dist.get_backend() if use_dist else None),
                       "dist_world_size": (dist.get_world_size() if use_dist else None),
                       "chain_streams_checked": dev.stats().get("chain_checks"), "chain_streams_rejected": dev.stats().get("chain_rejects"),
                       "plan_path": plan.get("path"), "group": group, "streams": chains,
                       "tile_w": plan.get("tile_w"), "launches_per_step": launches,
                       "scratch_bytes": plan.get("scratch_bytes")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "measured_floors": floors, "isolated_kernel_us": isolated,
                         "kernel": "k_p1_1m + k_p2_1m (the launches of one fwa_plan_exec: pass 1 then pass 2 per group "
                                   "of transforms, alternating over the chains)",
                         "ms_per_exec_hip_events": ev_ms, "ms_min": min(step_ms_events),
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
            "cpu_baseline": cpu,
            "gpu_phase_s": gpu_phase_s,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
