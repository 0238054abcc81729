"""ctypes loader for oracle/libfwo_oracle.so (TEST INFRASTRUCTURE ONLY).

Each wrapper cites the reference lines its C body follows; see ref_fft.c.
Arrays are numpy complex64 (interleaved re,im little-endian f32 -- the wire
layout of reference src/lib.rs:10-15).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfwo_oracle.so")

__all__ = [
    "build", "gen_input", "forward_ref", "inverse_ref", "onlyinverse_ref",
    "normalize_ref", "dft_f64", "dft_f64_naive", "compare", "bench_forward",
    "max_threads", "SEED",
]

SEED = 0x5EED


def build(force=False):
    """Compile the C restatement with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "ref_fft.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "ref_fft.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libfwo_oracle.so"])
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        c32p = ctypes.c_void_p
        lib.fwo_gen_input.argtypes = [c32p, ctypes.c_uint64, ctypes.c_uint64,
                                      ctypes.c_uint64, ctypes.c_uint32, ctypes.c_float]
        lib.fwo_gen_input.restype = None
        for name in ("fwo_forward_ref", "fwo_inverse_ref", "fwo_onlyinverse_ref"):
            f = getattr(lib, name)
            f.argtypes = [c32p, c32p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int]
            f.restype = ctypes.c_int
        lib.fwo_normalize_ref.argtypes = [c32p, c32p, ctypes.c_uint32, ctypes.c_uint64]
        lib.fwo_normalize_ref.restype = None
        lib.fwo_dft_f64.argtypes = [c32p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64,
                                    ctypes.c_int, ctypes.c_int]
        lib.fwo_dft_f64.restype = None
        lib.fwo_dft_f64_naive.argtypes = [c32p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
        lib.fwo_dft_f64_naive.restype = None
        lib.fwo_compare.argtypes = [c32p, ctypes.c_void_p, ctypes.c_uint32,
                                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.fwo_compare.restype = None
        lib.fwo_bench_forward.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int,
                                          ctypes.c_double, ctypes.POINTER(ctypes.c_int)]
        lib.fwo_bench_forward.restype = ctypes.c_double
        lib.fwo_max_threads.argtypes = []
        lib.fwo_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def max_threads():
    return int(_load().fwo_max_threads())


def gen_input(n, n_transforms, first_transform=0, seed=SEED, scale=1.0):
    """Deterministic uniform(-1,1) input, identical to the device generator."""
    out = np.empty(n * n_transforms, dtype=np.complex64)
    _load().fwo_gen_input(_ptr(out), seed, first_transform, n_transforms, n, scale)
    return out


def _run(fn, x, n, threads):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    assert x.size % n == 0
    a = x.copy()
    b = np.zeros_like(a)
    which = fn(_ptr(a), _ptr(b), n, x.size // n, threads or max_threads())
    return (a, b)[which], which


def forward_ref(x, n, threads=0):
    """fft4.wgsl:53-91 / fft.wgsl:27-62 with processor.rs:43-49 twiddles.
    Returns (result, which_buffer) -- which_buffer per processor.rs:153-157."""
    return _run(_load().fwo_forward_ref, x, n, threads)


def inverse_ref(x, n, threads=0):
    """ifft.wgsl:25-75 (1/n fused in the last stage)."""
    return _run(_load().fwo_inverse_ref, x, n, threads)


def onlyinverse_ref(x, n, threads=0):
    """onlyifft.wgsl:25-65 (no scale)."""
    return _run(_load().fwo_onlyinverse_ref, x, n, threads)


def normalize_ref(x, n):
    """normalize.wgsl:9-12."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty_like(x)
    _load().fwo_normalize_ref(_ptr(x), _ptr(out), n, x.size)
    return out


def dft_f64(x, n, direction=-1, threads=0):
    """Independent fp64 DFT of complex64 input; direction -1 forward, +1 inverse (unscaled)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty(x.size, dtype=np.complex128)
    _load().fwo_dft_f64(_ptr(x), _ptr(out), n, x.size // n, direction, threads or max_threads())
    return out


def dft_f64_naive(x, direction=-1):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty(x.size, dtype=np.complex128)
    _load().fwo_dft_f64_naive(_ptr(x), _ptr(out), x.size, direction)
    return out


def compare(y, r):
    """(max_k|y-r| / max_k|r|, rel-L2) for ONE transform; SURVEY.md 8(c) parity metric."""
    y = np.ascontiguousarray(y, dtype=np.complex64)
    r = np.ascontiguousarray(r, dtype=np.complex128)
    assert y.size == r.size
    a, b = ctypes.c_double(), ctypes.c_double()
    _load().fwo_compare(_ptr(y), _ptr(r), y.size, ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def bench_forward(n, batch, threads=0, min_seconds=10.0):
    """CPU baseline (kind 'port'): samples/s of the restatement, and repetitions run."""
    reps = ctypes.c_int(0)
    sps = _load().fwo_bench_forward(n, batch, threads or max_threads(), min_seconds,
                                    ctypes.byref(reps))
    return sps, reps.value
