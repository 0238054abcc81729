"""fft_wgpu_amd -- MI355X-native batched 1-D complex fp32 FFT behind the plan
API of the Rust crate TYPEmber/fft_wgpu (Forward / Inverse / Onlyinverse /
Normalize :: new, proc).  Hand-written HIP for gfx950 behind a C ABI
(include/fft_wgpu_amd.h); this package is the thin host-side mirror.
"""
from ._ffi import FwaError, LIB_PATH  # noqa: F401
from .device import (Buffer, CommandEncoder, Device, Event, Queue, device_count,  # noqa: F401
                     prepare_gpu)
from .processor import Forward, Inverse, Normalize, Onlyinverse  # noqa: F401

COMPLEX_BYTES = 8  # src/lib.rs:10-15: {real: f32, imag: f32}
