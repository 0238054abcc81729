"""Device / Queue / Buffer / CommandEncoder: the wgpu objects the reference's
callers touch (src/lib.rs:29-62, src/examples/basic.rs:6-122), mapped onto the
C ABI.  Names and call order follow the reference so that the example loop

    queue.write_buffer(src, 0, data); enc = device.create_command_encoder();
    out = plan.proc(enc); enc.copy_buffer_to_buffer(out, 0, staging, 0, n);
    queue.submit(enc.finish()); staging.map_read()

reads the same here.  Ordering follows the reference too: work recorded on an
encoder runs on that encoder's HIP stream; ``device.poll()`` (``Maintain::wait``,
examples/basic.rs:106) and a ``map_read()`` without a stream wait for ALL work
submitted to the device, so the literal sequence above never reads stale bytes.
"""
import ctypes

import numpy as np

from . import _ffi


class Buffer:
    """wgpu::Buffer (examples/basic.rs:50-64).  Caller-owned device memory."""

    def __init__(self, device, handle, borrowed=False):
        self.device = device
        self._h = handle
        self._borrowed = borrowed  # handle owned by a plan: never freed from here

    @property
    def size(self):
        return int(self.device._L.fwa_buf_size(self._h))

    @property
    def device_ptr(self):
        return int(self.device._L.fwa_buf_device_ptr(self._h) or 0)

    def destroy(self):
        if self._h and not self._borrowed:
            self.device._L.fwa_buf_free(self._h)
        self._h = None

    def map_read(self, offset=0, size=None, stream=None, dtype=np.complex64):
        """map_async + poll(wait) + get_mapped_range (examples/basic.rs:105-122): blocking read-back."""
        size = self.size - offset if size is None else size
        out = np.empty(size // np.dtype(dtype).itemsize, dtype=dtype)
        if stream is None:
            self.device.poll()  # map_async + poll(wait): everything submitted so far, on any encoder, has finished
        st = self.device._L.fwa_buf_download(out.ctypes.data_as(ctypes.c_void_p), self._h, offset, size,
                                         stream._h if stream else None)
        _ffi.check(st, self.device._h, "fwa_buf_download", self.device._L)
        return out

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class CommandEncoder:
    """wgpu::CommandEncoder: work recorded here runs in order on one HIP stream."""

    def __init__(self, device, stream_handle, owned):
        self.device = device
        self._h = stream_handle
        self._owned = owned

    def copy_buffer_to_buffer(self, src, src_offset, dst, dst_offset, size):
        st = self.device._L.fwa_buf_copy(dst._h, dst_offset, src._h, src_offset, size, self._h)
        _ffi.check(st, self.device._h, "fwa_buf_copy", self.device._L)

    def finish(self):
        return self

    def wait_for(self, other):
        """Device-side dependency: work recorded here after this call runs after everything on `other`."""
        _ffi.check(self.device._L.fwa_stream_wait_stream(self._h, other._h), self.device._h, "fwa_stream_wait_stream", self.device._L)

    def wait_event(self, event):
        """Device-side dependency on one recorded point (an Event) of another encoder."""
        _ffi.check(self.device._L.fwa_stream_wait_event(self._h, event._h), self.device._h, "fwa_stream_wait_event", self.device._L)

    def synchronize(self):
        _ffi.check(self.device._L.fwa_stream_synchronize(self._h), self.device._h, "fwa_stream_synchronize", self.device._L)

    def destroy(self):
        if self._h and self._owned:
            self.device._L.fwa_stream_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Event:
    def __init__(self, device):
        self.device = device
        h = ctypes.c_void_p()
        _ffi.check(self.device._L.fwa_event_create(device._h, ctypes.byref(h)), device._h, "fwa_event_create", self.device._L)
        self._h = h

    def record(self, encoder):
        _ffi.check(self.device._L.fwa_event_record(self._h, encoder._h), self.device._h, "fwa_event_record", self.device._L)

    def synchronize(self):
        _ffi.check(self.device._L.fwa_event_synchronize(self._h), self.device._h, "fwa_event_synchronize", self.device._L)

    def elapsed_ms(self, end):
        ms = ctypes.c_float()
        _ffi.check(self.device._L.fwa_event_elapsed_ms(self._h, end._h, ctypes.byref(ms)), self.device._h,
                   "fwa_event_elapsed_ms", self.device._L)
        return ms.value

    def __del__(self):
        try:
            if self._h:
                self.device._L.fwa_event_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Queue:
    """wgpu::Queue (examples/basic.rs:73,92)."""

    def __init__(self, device):
        self.device = device

    def write_buffer(self, buffer, offset, data, encoder=None):
        data = np.ascontiguousarray(data)
        st = self.device._L.fwa_buf_upload(buffer._h, offset, data.ctypes.data_as(ctypes.c_void_p), data.nbytes,
                                       encoder._h if encoder else None)
        _ffi.check(st, self.device._h, "fwa_buf_upload", self.device._L)
        if encoder is None:
            self.device.poll()

    def submit(self, command_buffer):
        # work was enqueued as it was recorded; submit is the ordering point the reference has
        return command_buffer


class Device:
    """wgpu::Device (+ Instance/Adapter): one per GPU ordinal."""

    def __init__(self, ordinal=0, lab=False):
        self._L = _ffi.lib(lab)  # lab=True: the laboratory build (tools/ and bit-identity tests only)
        self.lab = lab if isinstance(lab, str) else bool(lab)
        h = ctypes.c_void_p()
        _ffi.check(self._L.fwa_ctx_create(ordinal, ctypes.byref(h)), None, "fwa_ctx_create", self._L)
        self._h = h
        self.ordinal = ordinal
        d = ctypes.c_void_p()
        _ffi.check(self._L.fwa_stream_wrap(h, None, ctypes.byref(d)), h, "fwa_stream_wrap", self._L)
        self._default = CommandEncoder(self, d, owned=True)  # the HIP null stream

    def info(self):
        name = ctypes.create_string_buffer(256)
        cus = ctypes.c_int32()
        mem = ctypes.c_uint64()
        _ffi.check(self._L.fwa_ctx_device_info(self._h, name, 256, ctypes.byref(cus), ctypes.byref(mem)),
                   self._h, "fwa_ctx_device_info", self._L)
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}

    def create_buffer(self, size):
        h = ctypes.c_void_p()
        _ffi.check(self._L.fwa_buf_alloc(self._h, size, ctypes.byref(h)), self._h, "fwa_buf_alloc", self._L)
        return Buffer(self, h)

    def wrap_buffer(self, device_ptr, size):
        """Zero-copy view of memory owned elsewhere (e.g. a torch tensor's data_ptr())."""
        h = ctypes.c_void_p()
        _ffi.check(self._L.fwa_buf_wrap(self._h, device_ptr, size, ctypes.byref(h)), self._h, "fwa_buf_wrap", self._L)
        return Buffer(self, h)  # the handle is ours, the memory is not (fwa_buf_wrap)

    def create_command_encoder(self, hip_stream=None):
        h = ctypes.c_void_p()
        if hip_stream is None:
            _ffi.check(self._L.fwa_stream_create(self._h, ctypes.byref(h)), self._h, "fwa_stream_create", self._L)
            return CommandEncoder(self, h, owned=True)
        _ffi.check(self._L.fwa_stream_wrap(self._h, hip_stream, ctypes.byref(h)), self._h, "fwa_stream_wrap", self._L)
        return CommandEncoder(self, h, owned=True)

    def poll(self, encoder=None):
        """device.poll(Maintain::wait()) (examples/basic.rs:106): all submitted work (or one encoder's)."""
        if encoder is not None:
            encoder.synchronize()
        else:
            _ffi.check(self._L.fwa_ctx_synchronize(self._h), self._h, "fwa_ctx_synchronize", self._L)

    def stats(self):
        """Plan-cache counters of this context (fwa_ctx_get_i64)."""
        out = {}
        for k in ("table_builds", "table_cache_hits", "ring_allocs", "ring_reuses", "last_plan_create_us", "pooled_ring_bytes",
                  "mem_free_bytes", "mem_total_bytes", "chain_streams", "chain_checks", "chain_rejects", "chain_single_us",
                  "chain_pair_us"):
            v = ctypes.c_int64()
            _ffi.check(self._L.fwa_ctx_get_i64(self._h, k.encode(), ctypes.byref(v)), self._h, "fwa_ctx_get_i64", self._L)
            out[k] = v.value
        return out

    def set(self, key, value):
        """fwa_ctx_set_i64: "chain_check" = 0 turns the stream-overlap check of this context off."""
        _ffi.check(self._L.fwa_ctx_set_i64(self._h, key.encode(), int(value)), self._h, "fwa_ctx_set_i64", self._L)

    def get(self, key):
        v = ctypes.c_int64()
        _ffi.check(self._L.fwa_ctx_get_i64(self._h, key.encode(), ctypes.byref(v)), self._h, "fwa_ctx_get_i64", self._L)
        return v.value

    def peer_access(self, other):
        """0: the two devices cannot reach each other, 1: same device, 2: peer access (enabled by this call)."""
        k = ctypes.c_int32()
        _ffi.check(self._L.fwa_ctx_peer_access(self._h, other._h, ctypes.byref(k)), self._h, "fwa_ctx_peer_access", self._L)
        return k.value

    def pinned_array(self, n_elements, dtype=np.complex64):
        """Page-locked host staging array (the reference's MAP_READ staging buffer, examples/basic.rs:50-55)."""
        nbytes = int(n_elements) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        _ffi.check(self._L.fwa_host_alloc(self._h, nbytes, ctypes.byref(p)), self._h, "fwa_host_alloc", self._L)
        arr = np.ctypeslib.as_array((ctypes.c_char * nbytes).from_address(p.value)).view(dtype)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return arr

    def download_async(self, host_array, buffer, encoder, offset=0):
        st = self._L.fwa_buf_download_async(host_array.ctypes.data_as(ctypes.c_void_p), buffer._h, offset,
                                               host_array.nbytes, encoder._h)
        _ffi.check(st, self._h, "fwa_buf_download_async", self._L)

    def fill_synthetic(self, buffer, fft_len, seed=0x5EED, first_transform=0, scale=1.0, encoder=None):
        st = self._L.fwa_fill_synthetic(buffer._h, seed, first_transform, fft_len, scale,
                                           encoder._h if encoder else None)
        _ffi.check(st, self._h, "fwa_fill_synthetic", self._L)

    def calib_copy(self, dst, src, nbytes, encoder=None):
        st = self._L.fwa_calib_copy(dst._h, src._h, nbytes, encoder._h if encoder else None)
        _ffi.check(st, self._h, "fwa_calib_copy", self._L)

    def destroy(self):
        if self._h:
            self._L.fwa_ctx_destroy(self._h)
            self._h = None


def device_count():
    n = ctypes.c_int32()
    st = _ffi.lib().fwa_device_count(ctypes.byref(n))
    return n.value if st == 0 else 0


def enumerate_adapters():
    """instance.enumerate_adapters(..) (src/lib.rs:33-35): one entry per visible device ordinal, without creating a
    context: {"ordinal", "name", "compute_units", "hbm_bytes", "usable"} (usable = Device(ordinal) would succeed)."""
    L = _ffi.lib()
    out = []
    for o in range(device_count()):
        name = ctypes.create_string_buffer(256)
        cus, mem, ok = ctypes.c_int32(), ctypes.c_uint64(), ctypes.c_int32()
        _ffi.check(L.fwa_device_info(o, name, 256, ctypes.byref(cus), ctypes.byref(mem), ctypes.byref(ok)), None,
                   "fwa_device_info", L)
        out.append({"ordinal": o, "name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value,
                    "usable": bool(ok.value)})
    return out


def prepare_gpu(ordinal=0, lab=False):
    """src/lib.rs:29-62: returns (device, queue) or None when no adapter/device is usable."""
    try:
        dev = Device(ordinal, lab=lab)
    except _ffi.FwaError as e:
        if e.status == 5:  # FWA_ERR_NO_DEVICE
            return None
        raise
    return dev, Queue(dev)
