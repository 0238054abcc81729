"""fft_wgpu_amd -- MI355X-native batched 1-D complex fp32 FFT behind the plan
API of the Rust crate TYPEmber/fft_wgpu (Forward / Inverse / Onlyinverse /
Normalize :: new, proc).  Hand-written HIP for gfx950 behind a C ABI
(include/fft_wgpu_amd.h); this package is the thin host-side mirror.
"""
from ._ffi import FwaError, LIB_PATH  # noqa: F401
from .device import (Buffer, CommandEncoder, Device, Event, Queue, device_count,  # noqa: F401
                     enumerate_adapters, prepare_gpu)
from .pipeline import HostPipeline  # noqa: F401
from .processor import Forward, Inverse, Normalize, Onlyinverse  # noqa: F401
from .sharding import Comm, ShardedBatch, slab  # noqa: F401

COMPLEX_BYTES = 8  # src/lib.rs:10-15: {real: f32, imag: f32}


def describe_path(fft_len):
    """(path id, [per-pass FFT lengths]) a plan of this length will use -- pure host logic, no GPU needed."""
    import ctypes
    from . import _ffi
    path = ctypes.c_int32()
    lf = (ctypes.c_uint32 * 3)()
    _ffi.check(_ffi.lib().fwa_describe_path(fft_len, ctypes.byref(path), ctypes.byref(lf)), None, "fwa_describe_path")
    return path.value, [1 << v for v in lf if v]
