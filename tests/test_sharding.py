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
