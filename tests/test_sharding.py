"""not-gpu: the N>1 path.  Slab arithmetic, and world_size-2 gloo processes that shard a batch, run
each slab independently (CPU oracle standing in for the per-rank transform: the checker, not the
product) and gather -- the result must equal the unsharded transform bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_slabs_cover_batch_exactly():
    from fft_wgpu_amd.sharding import slab, slab_sizes
    for batch in (0, 1, 7, 4096, 32768, 32769):
        for world in (1, 2, 3, 4, 8):
            edges = [slab(batch, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and b >= a
            sizes = slab_sizes(batch, world)
            assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1
    assert slab(32768, 3, 8) == (3 * 4096, 4 * 4096)  # config C4: 4096 transforms per GPU
    with pytest.raises(ValueError):
        slab(8, 2, 2)


def test_c_abi_and_cpp_slab_rule_for_world_1_to_8(tmp_path):
    """one slab rule everywhere.  fwa_slab (the C ABI; sharding.slab calls it), fft_wgpu::slab
    (include/fft_wgpu.hpp, compiled here) and the divmod formula restated in this test agree for world 1..8, ragged and
    empty batches, C4's 32768 and counts beyond 2^32."""
    import ctypes
    import subprocess
    from fft_wgpu_amd import _ffi
    from fft_wgpu_amd.sharding import slab
    L = _ffi.lib()
    batches = [0, 1, 2, 5, 7, 8, 9, 63, 4096, 32768, 32769, (1 << 33) + 5]

    def formula(batch, r, world):
        base, extra = divmod(batch, world)
        first = r * base + min(r, extra)
        return first, base + (1 if r < extra else 0)

    src = tmp_path / "slabs.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "fft_wgpu.hpp"\n'
                   'int main(int argc, char **argv) {\n'
                   '  for (int i = 1; i < argc; ++i) for (int w = 1; w <= 8; ++w) for (int r = 0; r < w; ++r) {\n'
                   '    fft_wgpu::Slab s = fft_wgpu::slab(std::strtoull(argv[i], nullptr, 10), r, w);\n'
                   '    std::printf("%s %d %d %llu %llu\\n", argv[i], w, r, (unsigned long long)s.first, (unsigned long long)s.count); }\n'
                   '  try { fft_wgpu::slab(8, 2, 2); } catch (const fft_wgpu::Error &e) { std::printf("rejected %d\\n", e.status); }\n'
                   '  return 0; }\n')
    exe = tmp_path / "slabs"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), str(src),
                           "-L" + os.path.join(ROOT, "fft_wgpu_amd"), "-lfft_wgpu_amd", "-pthread", "-o", str(exe)])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "fft_wgpu_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)] + [str(b) for b in batches], env=env, capture_output=True, text=True, check=True).stdout.split("\n")
    cpp = {}
    for line in out:
        f = line.split()
        if len(f) == 5:
            cpp[(int(f[0]), int(f[1]), int(f[2]))] = (int(f[3]), int(f[4]))
    assert "rejected 1" in out                                   # FWA_ERR_INVALID_ARG, as sharding.slab's ValueError
    for batch in batches:
        for world in range(1, 9):
            nxt = 0
            for r in range(world):
                first, count = ctypes.c_uint64(), ctypes.c_uint64()
                assert L.fwa_slab(batch, r, world, ctypes.byref(first), ctypes.byref(count)) == 0
                want = formula(batch, r, world)
                assert (first.value, count.value) == want == cpp[(batch, world, r)]
                assert slab(batch, r, world) == (want[0], want[0] + want[1])
                assert first.value == nxt
                nxt += count.value
            assert nxt == batch
    f, c = ctypes.c_uint64(), ctypes.c_uint64()
    assert L.fwa_slab(8, 2, 2, ctypes.byref(f), ctypes.byref(c)) == 1 and L.fwa_slab(8, 0, 0, ctypes.byref(f), ctypes.byref(c)) == 1
    assert L.fwa_slab(8, 0, 1, None, ctypes.byref(c)) == 1


def test_multi_gpu_entry_points_reject_bad_handles_without_a_device():
    """The multi-GPU part of the boundary fails with status codes, never a crash, when there is nothing to run on."""
    import ctypes
    from fft_wgpu_amd import _ffi
    L = _ffi.lib()
    out = ctypes.c_void_p()
    k = ctypes.c_int32()
    assert L.fwa_comm_create(None, None, 1, 0, ctypes.byref(out)) == 1 and not out.value
    assert L.fwa_comm_scatter(None, 0, None, None, 1024, 4, None) == 1
    assert L.fwa_comm_gather(None, 0, None, None, 1024, 4, None) == 1
    assert L.fwa_comm_sendrecv(None, None, 0, 0, -1, None, 0, 0, -1, None) == 1
    assert L.fwa_comm_unique_id(None) == 1
    assert L.fwa_comm_destroy(None) == 0
    assert L.fwa_ctx_peer_access(None, None, ctypes.byref(k)) == 1
    assert L.fwa_ctx_set_i64(None, b"chain_check", 0) == 1
    assert L.fwa_buf_copy(None, 0, None, 0, 0, None) == 1
    if torch.cuda.device_count() == 0:
        import fft_wgpu_amd as fw
        assert fw.enumerate_adapters() == [] and fw.device_count() == 0
        ok = ctypes.c_int32(7)
        assert L.fwa_device_info(0, None, 0, None, None, ctypes.byref(ok)) == 5 and ok.value == 0   # FWA_ERR_NO_DEVICE
        with pytest.raises(fw.FwaError) as e:
            fw.ShardedBatch(fw.Forward, 1024, 8)
        assert e.value.status == 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, batch, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from fft_wgpu_amd.sharding import gather_batch, scatter_batch, slab
    full = None
    if rank == 0:
        x = oracle.gen_input(n, batch)
        full = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2).copy())
    mine = scatter_batch(full, n, src=0)
    lo, hi = slab(batch, rank, world)
    assert mine.shape[0] == hi - lo
    # each rank's slab equals what the device generator would produce in place (bench.py path)
    local = oracle.gen_input(n, hi - lo, first_transform=lo)
    assert np.array_equal(mine.numpy().reshape(-1).view(np.complex64), local)
    y, _ = oracle.forward_ref(local, n, threads=1)
    out = gather_batch(torch.from_numpy(y.view(np.float32).reshape(hi - lo, n, 2).copy()), batch, n, dst=0)
    if rank == 0:
        ref, _ = oracle.forward_ref(oracle.gen_input(n, batch), n, threads=1)
        q.put(bool(np.array_equal(out.numpy().reshape(-1).view(np.complex64).view(np.uint32), ref.view(np.uint32))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [5, 8])
def test_world_size_2_gloo_scatter_transform_gather(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 256, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher starts its own ranks; with fewer than N devices visible it must
    exit non-zero instead of reporting an N-GPU figure from fewer GPUs (here: no GPU at all)."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert '"n_gpus"' not in r.stdout           # no benchmark line was produced
    # a launcher environment that disagrees with --gpus is rejected as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_rank_slab_is_sharding_slab():
    """bench.py gives rank r the transforms sharding.slab(batch * world, r, world): config C4 = 4096 per GPU x 8."""
    from fft_wgpu_amd.sharding import slab
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "sharding.slab(batch * world, rank, world)" in src
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert slab(4096 * world, r, world) == (r * 4096, (r + 1) * 4096)


def test_bench_launcher_stops_the_other_ranks_when_one_dies():
    """bench.py's own launcher (`--gpus N` without a launcher environment) supervises its ranks: the first one that exits
    non-zero ends the run, siblings get SIGTERM and -- if they ignore it -- SIGKILL, and its status is returned; ranks
    that all exit 0 give 0.  Stand-in children here (no GPU); the real thing runs in tests/test_gpu_bench.py."""
    import io
    import subprocess
    import time
    sys.path.insert(0, ROOT)
    import bench
    sleeper = "import time; time.sleep(120)"
    stubborn = "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(120)"
    procs = [subprocess.Popen([sys.executable, "-c", sleeper]),
             subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"]),
             subprocess.Popen([sys.executable, "-c", stubborn])]
    log = io.StringIO()
    t0 = time.monotonic()
    rc = bench.supervise(procs, grace_s=1.0, log=log)
    assert rc == 3 and time.monotonic() - t0 < 20.0
    assert [p.returncode for p in procs] == [-15, 3, -9]
    assert "rank 1 exited with status 3" in log.getvalue()
    assert bench.supervise([subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]) == 0
    killed = subprocess.Popen([sys.executable, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGKILL)"])
    assert bench.supervise([killed], log=io.StringIO()) == 128 + 9


def test_scatter_gather_piece_tables_for_world_2_to_8():
    """the per-peer tables fwa_comm_scatter / fwa_comm_gather post (fwa_comm_pieces: pure host logic, the
    function both collectives build their grouped sends / receives from) for every world size 1..8, every root, ragged
    and empty batches: the root's pieces tile the full batch in rank order with peer = rank and -1 for its own slab; every
    other rank posts exactly one piece, its slab, to / from the root.  What stays untested without a second GPU is the
    RCCL call itself."""
    import ctypes
    from fft_wgpu_amd import _ffi
    from fft_wgpu_amd.sharding import slab
    L = _ffi.lib()
    for n in (512, 1 << 20):
        tb = 8 * n
        for world in range(1, 9):
            off, nb, peer = (ctypes.c_uint64 * world)(), (ctypes.c_uint64 * world)(), (ctypes.c_int32 * world)()
            cnt = ctypes.c_int32()
            for batch in (0, 1, world - 1, world, 3 * world + 2, 4096 * world, (1 << 33) + 5):
                for root in range(world):
                    for rank in range(world):
                        assert L.fwa_comm_pieces(batch, n, root, rank, world, off, nb, peer, ctypes.byref(cnt)) == 0
                        lo, hi = slab(batch, rank, world)
                        if rank != root:
                            assert cnt.value == 1 and (off[0], nb[0], peer[0]) == (0, (hi - lo) * tb, root)
                            continue
                        assert cnt.value == world
                        nxt = 0
                        for p in range(world):
                            a, b = slab(batch, p, world)
                            assert (off[p], nb[p]) == (a * tb, (b - a) * tb) and off[p] == nxt
                            assert peer[p] == (-1 if p == root else p)
                            nxt += nb[p]
                        assert nxt == batch * tb
    cnt = ctypes.c_int32(9)
    one = (ctypes.c_uint64 * 1)()
    pr = (ctypes.c_int32 * 1)()
    assert L.fwa_comm_pieces(8, 512, 2, 0, 2, one, one, pr, ctypes.byref(cnt)) == 1 and cnt.value == 0   # root out of range
    assert L.fwa_comm_pieces(8, 0, 0, 0, 1, one, one, pr, ctypes.byref(cnt)) == 1                       # fft_len 0
    assert L.fwa_comm_pieces(8, 512, 0, 0, 1, None, one, pr, ctypes.byref(cnt)) == 1
