"""Batch sharding across the GPUs of one node.

Every length-n transform depends only on its own n samples (reference
src/kernel/fft4.wgsl:21-23: one `offset` per workgroup), so a batch shards as
contiguous slabs of whole transforms, one process per GPU, with NO data-path
collective.  RCCL (torch.distributed backend "nccl"; "gloo" in the CPU tests)
is used only by the optional slab scatter / gather below, for callers whose
data starts on one rank.
"""
import torch
import torch.distributed as dist


def slab(batch, rank, world_size):
    """[first, last) transform indices of `rank`'s slab; slabs differ by at most one transform."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(batch, world_size)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def slab_sizes(batch, world_size):
    return [slab(batch, r, world_size)[1] - slab(batch, r, world_size)[0] for r in range(world_size)]


def scatter_batch(full, fft_len, src=0, group=None):
    """Rank `src` holds `full` (float32 tensor viewed as [batch, fft_len, 2]); every rank returns its slab.

    Whole transforms only; implemented as point-to-point sends (xGMI is point-to-point: SURVEY.md 5)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = torch.zeros(1, dtype=torch.int64)
    if rank == src:
        assert full.dim() == 3 and full.shape[1] == fft_len and full.shape[2] == 2
        meta[0] = full.shape[0]
    dev = full.device if (rank == src) else None
    if dist.get_backend(group) == "nccl":
        meta = meta.cuda()
    dist.broadcast(meta, src, group=group)
    batch = int(meta.item())
    lo, hi = slab(batch, rank, world)
    if rank == src:
        reqs = []
        for r in range(world):
            if r == src:
                continue
            a, b = slab(batch, r, world)
            if b > a:
                reqs.append(dist.isend(full[a:b].contiguous(), r, group=group))
        mine = full[lo:hi].clone()
        for q in reqs:
            q.wait()
        return mine
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.empty((hi - lo, fft_len, 2), dtype=torch.float32, device=device if dev is None else dev)
    if hi > lo:
        dist.recv(mine, src, group=group)
    return mine


def gather_batch(mine, batch, fft_len, dst=0, group=None):
    """Inverse of scatter_batch: rank `dst` returns the [batch, fft_len, 2] tensor, the others None."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank != dst:
        if mine.shape[0]:
            dist.send(mine.contiguous(), dst, group=group)
        return None
    full = torch.empty((batch, fft_len, 2), dtype=torch.float32, device=mine.device)
    for r in range(world):
        a, b = slab(batch, r, world)
        if r == dst:
            full[a:b] = mine
        elif b > a:
            dist.recv(full[a:b], r, group=group)
    return full
