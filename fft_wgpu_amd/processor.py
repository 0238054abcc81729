"""The four plan objects of reference src/processor.rs, same names and call
shape (``X(device, queue, src[, src2], fft_len)``, ``X.proc(encoder) -> Buffer``),
executing hand-written HIP through the C ABI.
"""
import ctypes

from . import _ffi
from .device import Buffer


# Tuning keys of fwa_plan_set_i64 in the order they must be applied ("factors" re-plans the pipeline, so it comes
# before "group" / "streams").  No key changes what a plan computes.
PLAN_KEYS = ("path", "factors", "group", "streams", "tile_w", "xcd_swizzle", "depth", "ring_slots", "wgs",
             "small_reg", "p1_gen", "rows32", "colsw", "tile_ring", "ring_rotate", "inject_launch_failure")


class _Plan:
    _kind = None

    def __init__(self, device, queue, src, src2, fft_len):
        self.device = device
        self.queue = queue          # kept for signature parity; unused, as in the reference (processor.rs:9)
        self.buffer_a = src
        self.buffer_b = src2
        self.fft_len = fft_len
        h = ctypes.c_void_p()
        st = self.device._L.fwa_plan_create(device._h, self._kind, fft_len, src._h,
                                        src2._h if src2 is not None else None, ctypes.byref(h))
        _ffi.check(st, device._h, f"fwa_plan_create({type(self).__name__})", self.device._L)
        self._h = h
        self._results = {}

    @classmethod
    def new(cls, *args):
        return cls(*args)

    def proc(self, encoder):
        """Enqueue the transform on ``encoder`` and return the buffer that will hold the result."""
        res = ctypes.c_void_p()
        st = self.device._L.fwa_plan_exec(self._h, encoder._h if encoder is not None else None, ctypes.byref(res))
        _ffi.check(st, self.device._h, "fwa_plan_exec", self.device._L)
        for b in (self.buffer_a, self.buffer_b):
            if b is not None and b._h is not None and b._h.value == res.value:
                return b
        # plan-owned second buffer (Forward/Inverse with odd log2 n): a borrowed view that keeps the plan
        # alive (the Rust API ties it to the plan's lifetime: `proc(&self) -> &wgpu::Buffer`)
        view = Buffer(self.device, ctypes.c_void_p(res.value), borrowed=True)
        view._plan = self
        return view

    def get(self, key):
        v = ctypes.c_int64()
        _ffi.check(self.device._L.fwa_plan_get_i64(self._h, key.encode(), ctypes.byref(v)), self.device._h,
                   "fwa_plan_get_i64", self.device._L)
        return v.value

    def set(self, key, value):
        _ffi.check(self.device._L.fwa_plan_set_i64(self._h, key.encode(), int(value)), self.device._h,
                   "fwa_plan_set_i64", self.device._L)

    def destroy(self):
        if self._h:
            self.device._L.fwa_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Forward(_Plan):
    """processor.rs:7-159 + kernel/fft4.wgsl: unnormalised forward DFT."""
    _kind = _ffi.FORWARD

    def __init__(self, device, queue, src, fft_len):
        super().__init__(device, queue, src, None, fft_len)


class Inverse(_Plan):
    """processor.rs:231-341 + kernel/ifft.wgsl: inverse DFT with the 1/n scale fused."""
    _kind = _ffi.INVERSE_SCALED

    def __init__(self, device, queue, src, fft_len):
        super().__init__(device, queue, src, None, fft_len)


class Onlyinverse(_Plan):
    """processor.rs:566-670 + kernel/onlyifft.wgsl: inverse DFT, unscaled, caller-supplied second buffer."""
    _kind = _ffi.INVERSE_UNSCALED

    def __init__(self, device, queue, src, src2, fft_len):
        super().__init__(device, queue, src, src2, fft_len)


class Normalize(_Plan):
    """processor.rs:409-505 + kernel/normalize.wgsl: b[i] = a[i] / fft_len."""
    _kind = _ffi.NORMALIZE

    def __init__(self, device, queue, buffer1, buffer2, fft_len):
        super().__init__(device, queue, buffer1, buffer2, fft_len)
