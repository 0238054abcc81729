"""Batch sharding across the GPUs of one node.

Every length-n transform depends only on its own n samples (reference
src/kernel/fft4.wgsl:21-23: one `offset` per workgroup), so a batch shards as
contiguous slabs of whole transforms, one slab per GPU, with NO data-path
collective.  Three ways to drive it, all on the slab rule of the C ABI
(`fwa_slab`):

* one process per GPU over ``torch.distributed`` (backend "nccl" = RCCL; "gloo" in the CPU
  tests): ``bench.py``, ``scatter_batch`` / ``gather_batch`` below;
* one process per GPU over the C ABI alone: ``Comm`` (``fwa_comm_*``: RCCL loaded by the
  library itself, grouped send / receive), for hosts without torch -- the Rust crate, C++;
* ONE process driving every GPU: ``ShardedBatch`` (one context + encoder + plan per ordinal,
  slabs moved with peer copies, ``fwa_buf_copy`` across contexts) -- the Python twin of
  ``fft_wgpu::ShardedBatch`` in include/fft_wgpu.hpp.

Data movement (scatter / gather) is never part of a throughput figure: a root pushing
7 x 32 GiB over xGMI needs ~0.2 s against ~21 ms of transform.
"""
import ctypes

from . import _ffi


def slab(batch, rank, world_size):
    """[first, last) transform indices of `rank`'s slab; slabs differ by at most one transform (fwa_slab)."""
    first, count = ctypes.c_uint64(), ctypes.c_uint64()
    st = _ffi.lib().fwa_slab(int(batch), int(rank), int(world_size), ctypes.byref(first), ctypes.byref(count))
    if st:
        raise ValueError("bad rank/world_size")
    return first.value, first.value + count.value


def slab_sizes(batch, world_size):
    return [slab(batch, r, world_size)[1] - slab(batch, r, world_size)[0] for r in range(world_size)]


def scatter_batch(full, fft_len, src=0, group=None):
    """Rank `src` holds `full` (float32 tensor viewed as [batch, fft_len, 2]); every rank returns its slab.

    Whole transforms only; implemented as point-to-point sends (xGMI is point-to-point: SURVEY.md 5)."""
    import torch
    import torch.distributed as dist
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
    import torch
    import torch.distributed as dist
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


class Comm:
    """fwa_comm_*: one rank of a slab communicator over RCCL, for one-process-per-GPU hosts that do not go through
    torch.distributed.  ``Comm.unique_id()`` on rank 0, hand the 128 bytes to the other ranks, then
    ``Comm(device, id, world, rank)`` on every rank (collective)."""

    @staticmethod
    def unique_id(L=None):
        L = L or _ffi.lib()
        buf = ctypes.create_string_buffer(128)
        _ffi.check(L.fwa_comm_unique_id(buf), None, "fwa_comm_unique_id", L)
        return buf.raw

    def __init__(self, device, unique_id, world, rank):
        self.device = device
        h = ctypes.c_void_p()
        _ffi.check(device._L.fwa_comm_create(device._h, unique_id, world, rank, ctypes.byref(h)), device._h,
                   "fwa_comm_create", device._L)
        self._h = h
        self.world, self.rank = world, rank

    def get(self, key):
        v = ctypes.c_int64()
        _ffi.check(self.device._L.fwa_comm_get_i64(self._h, key.encode(), ctypes.byref(v)), self.device._h, "fwa_comm_get_i64",
                   self.device._L)
        return v.value

    def scatter(self, full, slab_buf, fft_len, batch, root=0, encoder=None):
        """root's `full` (None elsewhere) -> every rank's `slab_buf`, stream-ordered on `encoder`."""
        st = self.device._L.fwa_comm_scatter(self._h, root, full._h if full is not None else None, slab_buf._h, fft_len, batch,
                                             encoder._h if encoder else None)
        _ffi.check(st, self.device._h, "fwa_comm_scatter", self.device._L)

    def gather(self, slab_buf, full, fft_len, batch, root=0, encoder=None):
        st = self.device._L.fwa_comm_gather(self._h, root, slab_buf._h, full._h if full is not None else None, fft_len, batch,
                                            encoder._h if encoder else None)
        _ffi.check(st, self.device._h, "fwa_comm_gather", self.device._L)

    def sendrecv(self, send, send_offset, send_bytes, send_to, recv, recv_offset, recv_bytes, recv_from, encoder=None):
        st = self.device._L.fwa_comm_sendrecv(self._h, send._h if send is not None else None, send_offset, send_bytes, send_to,
                                              recv._h if recv is not None else None, recv_offset, recv_bytes, recv_from,
                                              encoder._h if encoder else None)
        _ffi.check(st, self.device._h, "fwa_comm_sendrecv", self.device._L)

    def destroy(self):
        if self._h:
            self.device._L.fwa_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ShardedBatch:
    """One batch of `batch` transforms sharded over several devices driven by THIS process: shard i = one Device +
    CommandEncoder + slab buffer + plan on ordinal ``ordinals[i]``, slab i = ``slab(batch, i, len(ordinals))``.
    ``proc()`` enqueues every shard's transform and returns the result buffers (the reference's ``proc`` returns the
    buffer that holds the result, src/processor.rs:153-157); ``poll()`` joins.  ``scatter`` / ``gather`` move slabs from /
    to one buffer on one shard's device with peer copies (``fwa_buf_copy`` across contexts); ``write`` / ``read`` move
    them from / to host memory, every device taking only its own slab.

    ``ordinals`` defaults to every visible device; an ordinal may repeat (two contexts on one device: the degenerate form
    the one-GPU tests use).  ``devices``: use these Device objects instead of creating one per ordinal -- the caller
    keeps them (``destroy()`` then leaves them alone; devices created here are destroyed there).

    ``proc()`` enqueues the shards from one host thread per shard when an exec is many launches (a C3-sized slab is 512
    launches + 512 event operations = about 3 ms of host time per device against 21 ms of GPU work: serially, eight
    devices would be host-bound; profiles/round5/enqueue_cost.jsonl); ctypes releases the GIL during the call.
    ``threads``: None = by launch count, True / False = always / never."""

    THREAD_MIN_LAUNCHES = 16   # below this an exec returns in < 0.1 ms and a thread hand-off costs as much

    def __init__(self, plan_cls, fft_len, batch, ordinals=None, lab=False, devices=None, threads=None):
        from .device import Device, Queue, device_count
        if devices is not None:
            ordinals = [d.ordinal for d in devices]
        elif ordinals is None:
            ordinals = list(range(device_count()))
        if not ordinals:
            raise _ffi.FwaError(5, "no device visible", "ShardedBatch")
        self.fft_len, self.batch = fft_len, batch
        self._owns_devices = devices is None
        self.threads = threads
        self._pool = None
        self._many_launches = None
        self.devices = list(devices) if devices is not None else [Device(o, lab=lab) for o in ordinals]
        self.queues = [Queue(d) for d in self.devices]
        self.encoders = [d.create_command_encoder() for d in self.devices]
        world = len(ordinals)
        self.slabs = [slab(batch, r, world) for r in range(world)]
        tb = 8 * fft_len
        self.buffers = [d.create_buffer((b - a) * tb) for d, (a, b) in zip(self.devices, self.slabs)]
        needs_second = plan_cls.__name__ in ("Onlyinverse", "Normalize")
        self.seconds = [d.create_buffer((b - a) * tb) for d, (a, b) in zip(self.devices, self.slabs)] if needs_second else None
        if needs_second:
            self.plans = [plan_cls(d, q, b, s, fft_len) for d, q, b, s in zip(self.devices, self.queues, self.buffers, self.seconds)]
        else:
            self.plans = [plan_cls(d, q, b, fft_len) for d, q, b in zip(self.devices, self.queues, self.buffers)]
        self.results = list(self.buffers)

    def __len__(self):
        return len(self.devices)

    def _threaded(self):
        if self.threads is not None:
            return bool(self.threads) and len(self.plans) > 1
        if self._many_launches is None:      # the plans do not change after construction
            self._many_launches = max(p.get("launches_per_exec") for p in self.plans) >= self.THREAD_MIN_LAUNCHES
        return len(self.plans) > 1 and self._many_launches

    def proc(self):
        """Enqueue the transform of every slab on its device's encoder; returns the list of result buffers.  Returns when
        every shard's launches are queued (not when they have run): with one enqueueing thread per shard that is the
        host time of ONE shard, not their sum."""
        if self._threaded():
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=len(self.plans), thread_name_prefix="fwa-shard")
            from concurrent.futures import wait
            futures = [self._pool.submit(p.proc, e) for p, e in zip(self.plans, self.encoders)]
            # EVERY shard has finished enqueueing before the first failure is re-raised: a destroy() issued from the caller's
            # except block must not race with shards that are still inside fwa_plan_exec
            wait(futures)
            self.results = [f.result() for f in futures]      # re-raises a shard's FwaError here
        else:
            self.results = [p.proc(e) for p, e in zip(self.plans, self.encoders)]
        return self.results

    def poll(self):
        for e in self.encoders:
            e.synchronize()

    def write(self, data):
        """`data`: complex64 array of batch * fft_len samples in host memory; slab i goes to device i."""
        n = self.fft_len
        for q, b, e, (a, z) in zip(self.queues, self.buffers, self.encoders, self.slabs):
            if z > a:
                q.write_buffer(b, 0, data[a * n:z * n], encoder=e)
        self.poll()

    def read(self, out=None):
        import numpy as np
        n = self.fft_len
        out = np.empty(self.batch * n, dtype=np.complex64) if out is None else out
        for r, e, (a, z) in zip(self.results, self.encoders, self.slabs):
            if z > a:
                out[a * n:z * n] = r.map_read(size=(z - a) * n * 8, stream=e)
        return out

    def scatter(self, full, root=0):
        """`full`: a buffer of the whole batch on shard `root`'s device -> every shard's slab buffer (peer copies)."""
        tb = 8 * self.fft_len
        self.encoders[root].synchronize()
        for i, (b, e, (a, z)) in enumerate(zip(self.buffers, self.encoders, self.slabs)):
            if z > a:
                st = self.devices[i]._L.fwa_buf_copy(b._h, 0, full._h, a * tb, (z - a) * tb, e._h)
                _ffi.check(st, self.devices[i]._h, "fwa_buf_copy", self.devices[i]._L)

    def gather(self, full, root=0):
        """every shard's RESULT buffer -> `full` on shard `root`'s device, each copy on the producing shard's encoder."""
        tb = 8 * self.fft_len
        for i, (r, e, (a, z)) in enumerate(zip(self.results, self.encoders, self.slabs)):
            if z > a:
                st = self.devices[i]._L.fwa_buf_copy(full._h, a * tb, r._h, 0, (z - a) * tb, e._h)
                _ffi.check(st, self.devices[i]._h, "fwa_buf_copy", self.devices[i]._L)
        self.poll()

    def destroy(self):
        for p in self.plans:
            p.destroy()
        for group in (self.buffers, self.seconds or []):
            for b in group:
                b.destroy()
        for e in self.encoders:
            e.destroy()
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        # contexts this object created go last (each holds its pooled ring slab, twiddle tables and chain streams until then);
        # buffer, stream and event handles a caller still holds on them stay destroyable, and USING one afterwards is
        # FWA_ERR_INVALID_ARG, never a dangling pointer: fwa_ctx_destroy detaches the context's live handles
        # (include/fft_wgpu_amd.h, "Lifetimes").  Devices the caller passed in are the caller's.
        if self._owns_devices:
            for d in self.devices:
                d.destroy()
        self.plans, self.buffers, self.seconds, self.encoders, self.results = [], [], None, [], []
