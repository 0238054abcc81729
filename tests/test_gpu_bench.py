"""GPU tests of bench.py's N > 1 path on a one-GPU box.

The driver's SCALE job is the first time `bench.py --gpus N` runs on N real GPUs, so everything around the timed region
-- rendezvous, barriers, MAX / all-gather over ranks, rank-0-only printing, slab offsets, the launcher that supervises its
ranks -- is dress-rehearsed here with N ranks on device 0 (FWA_BENCH_REHEARSAL=1: "gloo" process group, the line marked
`"rehearsal": true`).  Every bench run is a FRESH child process (this process has initialised the GPU and must not
re-exec); independence of the slabs: reference src/kernel/fft4.wgsl:21-23.
"""
import json
import os
import socket
import subprocess
import sys
import time
import uuid

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FWA_BENCH_FORCE_DIST")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def _one_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-3000:]
    return json.loads(lines[0])


def _check_rehearsal_line(line, world, batch):
    assert line["n_gpus"] == world and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["rehearsal"] is True and line["vs_baseline"] is None
    assert line["config"]["dist_backend"] == "gloo" and line["config"]["dist_world_size"] == world
    # rank r owns transforms [r * batch, (r + 1) * batch) of the global batch: sharding.slab(batch * world, r, world)
    assert line["config"]["slab_first_transform_per_rank"] == [r * batch for r in range(world)]
    assert len(line["per_rank_ms"]) == world and len(line["per_rank_hip_event_ms"]) == world
    assert all(np.isfinite(v) and v > 0 for v in line["per_rank_ms"] + line["per_rank_hip_event_ms"])
    assert line["per_rank_ms_min"] == min(line["per_rank_ms"]) and line["per_rank_ms_max"] == max(line["per_rank_ms"])
    # MAX over ranks: the step time of the line is no shorter than any rank's own
    assert line["ms_per_step"] >= line["per_rank_ms_max"] * 0.999
    assert np.isfinite(line["value"]) and line["value"] > 0
    want = batch * world * line["config"]["fft_len"] / (line["ms_per_step"] * 1e-3) / 1e9
    assert abs(line["value"] - want) <= 1e-6 * want
    assert 0 <= line["roofline"]["rank"] < world and 0 < line["roofline"]["frac"] < 1.0
    assert line["roofline"]["ms_per_exec_hip_events"] == max(line["per_rank_hip_event_ms"])    # the slowest rank's GPU
    assert line["cpu_baseline"] and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert "skipped" in line["movement"]


def test_rehearsal_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2` with no launcher environment: bench.py starts and supervises its own two ranks."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256",
                        "--cpu-seconds", "1", "--spread", "0"],
                       env=_clean_env(FWA_BENCH_REHEARSAL="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    _check_rehearsal_line(_one_line(r.stdout), 2, 256)


def test_rehearsal_three_ranks_under_the_drivers_launcher_command():
    """The command the driver uses for N > 1 (torch.distributed.run, one rank per LOCAL_RANK), three ranks on device 0."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "3",
                        "--steps", "2", "--warmup", "1", "--batch", "128", "--cpu-seconds", "1", "--spread", "0"],
                       env=_clean_env(FWA_BENCH_REHEARSAL="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    _check_rehearsal_line(_one_line(r.stdout), 3, 128)


def _tagged_processes(tag):
    import psutil
    found = []
    for p in psutil.process_iter():
        try:
            if p.environ().get("FWA_BENCH_TEST_TAG") == tag:
                found.append(p.pid)
        except (psutil.Error, OSError):
            pass
    return found


def test_a_dead_rank_takes_the_run_down_quickly_and_leaves_no_process_behind():
    """Rank 1 exits with status 3 before the rendezvous (FWA_BENCH_FAIL_RANK): rank 0 would wait for it in
    init_process_group for 300 s; the supervising parent must stop it and return non-zero within 30 s."""
    tag = uuid.uuid4().hex
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline", "--spread", "0"],
                       env=_clean_env(FWA_BENCH_REHEARSAL="1", FWA_BENCH_FAIL_RANK="1", FWA_BENCH_TEST_TAG=tag),
                       capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert r.returncode == 3, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert took < 30.0, took
    assert "rank 1 exited with status 3" in r.stderr
    assert '"n_gpus"' not in r.stdout                      # no benchmark line from a broken run
    deadline = time.monotonic() + 5.0
    while _tagged_processes(tag) and time.monotonic() < deadline:
        time.sleep(0.2)
    assert _tagged_processes(tag) == []


def test_distributed_leg_on_rccl_with_the_movement_leg():
    """World size 1 with FWA_BENCH_FORCE_DIST=1: the "nccl" (= RCCL) process group, device barrier, MAX all-reduce,
    all-gather of the per-rank times and the fwa_comm_* movement leg (communicator from a broadcast unique id, scatter
    and gather timed with HIP events) all execute on real RCCL -- in their one-rank form, the only one a one-GPU box has."""
    env = _clean_env(FWA_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
                     LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--spread", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["rehearsal"] is False
    assert line["config"]["dist_backend"] == "nccl" and line["config"]["dist_world_size"] == 1
    assert np.isfinite(line["value"]) and line["value"] > 50.0, line["value"]          # Gsamples/s; ~200 on an MI355X
    assert 0.2 < line["roofline"]["frac"] < 1.0 and line["roofline"]["traffic_source"]
    assert line["cpu_baseline"] is None and line["per_rank_ms"] and len(line["per_rank_ms"]) == 1
    mv = line["movement"]
    assert "error" not in mv and "skipped" not in mv, mv
    assert mv["transforms_per_rank"] == 256 and mv["bytes_per_rank"] == 256 << 23
    assert mv["scatter_ms"] > 0 and mv["gather_ms"] > 0 and mv["scatter_GBps"] is None   # nothing leaves rank 0 at world 1


def test_a_failing_movement_leg_keeps_the_measured_line():
    """The slab-movement leg fails on a rank (FWA_BENCH_FAIL_MOVEMENT): no further collective is safe, so the rank prints the
    complete line with the failure recorded in `movement` and leaves with status 0 -- the timed figures stand."""
    env = _clean_env(FWA_BENCH_FORCE_DIST="1", FWA_BENCH_FAIL_MOVEMENT="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                     RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--batch", "256",
                        "--spread", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    line = _one_line(r.stdout)
    assert "on request" in line["movement"]["error"] and np.isfinite(line["value"]) and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["steps"] == 2


def test_default_line_carries_every_single_gpu_config_of_baseline_json():
    """C1's shape (1024 x 1), C2 (2^20 x 1) and C5 (2^24 x 1) are timed and checked INSIDE the
    driver-run command, in a `configs` object of the one JSON line (BASELINE.json configs[0, 1, 4]; reference
    src/processor.rs:110-158).  A fresh child with the headline batch cut down (the configs leg does not depend on it)."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--batch", "256", "--cpu-seconds", "1",
                        "--spread", "0", "--config-execs", "200"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    line = _one_line(r.stdout)
    cfg = line["configs"]
    assert set(cfg) == {"C1_shape", "C2", "C5", "_clock"}
    want = {"C1_shape": (1 << 10, 1, 1.0, 20.0), "C2": (1 << 20, 3, 8.0, 40.0), "C5": (1 << 24, 3, 60.0, 300.0)}
    for name, (n, launches, lo_us, hi_us) in want.items():
        c = cfg[name]
        assert c["fft_len"] == n and c["batch"] == 1 and c["execs"] == 200 and c["launches"] == launches, c
        assert lo_us < c["us_min"] <= c["us_median"] <= c["us_p90"] < hi_us, c
        assert lo_us < c["us_queued_min"] <= c["us_queued_median"] < hi_us, c
        assert abs(c["Gsamples_per_s"] - n / c["us_median"] / 1e3) <= 1e-6 * c["Gsamples_per_s"]
        assert abs(c["frac"] - 16 * n / (c["us_median"] * 1e-6) / 8e12) <= 1e-9 and 0 < c["frac"] < 1
        assert c["parity_ok"] is True and 0 < c["parity_max_rel"] <= 1e-5, c
    assert "oracle.dft_f64" in cfg["C2"]["parity_checker"] and "impulse" in cfg["C5"]["parity_checker"]
    # the headline is untouched by the extra leg: still the first-class fields of the contract
    assert line["metric"].startswith("Gsamples/s") and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["cores"] >= 1
