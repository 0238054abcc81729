"""ctypes binding of include/fft_wgpu_amd.h (the C-ABI drop-in boundary).

There is no fallback: if the HIP library is missing or a call fails, this
module raises.  Nothing here imports ``oracle``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfft_wgpu_amd.so")
# The laboratory build (`make -C fft_wgpu_amd/csrc lab`): the product library plus the kernel families that measured
# slower than the defaults and are still cited (the persistent 2^20 ring, the direct 16-point kernels with the wavefront-
# shuffle exchange) and two test knobs.  Same ABI; only tools/ and the bit-identity tests load it (Device(lab=True)).
LAB_LIB_PATH = os.environ.get("FWA_LAB_LIBRARY") or os.path.join(_HERE, "libfft_wgpu_amd_lab.so")  # the override is for tools/ A/B builds

FWA_OK = 0
ABI_VERSION = 4   # FWA_ABI_VERSION of the include/fft_wgpu_amd.h this binding was written against
FORWARD, INVERSE_SCALED, INVERSE_UNSCALED, NORMALIZE = 0, 1, 2, 3
STATUS_NAMES = {
    0: "FWA_OK", 1: "FWA_ERR_INVALID_ARG", 2: "FWA_ERR_OUT_OF_MEMORY", 3: "FWA_ERR_HIP",
    4: "FWA_ERR_LAUNCH", 5: "FWA_ERR_NO_DEVICE", 6: "FWA_ERR_UNSUPPORTED",
}


class FwaError(RuntimeError):
    def __init__(self, status, detail, where):
        self.status = status
        self.detail = detail
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}: {detail}")


_P = ctypes.c_void_p
_PP = ctypes.POINTER(ctypes.c_void_p)
_U64 = ctypes.c_uint64
_U32 = ctypes.c_uint32
_I32 = ctypes.c_int32

_SIGNATURES = {
    "fwa_abi_version": (_I32, []),
    "fwa_last_error_string": (ctypes.c_char_p, [_P]),
    "fwa_status_string": (ctypes.c_char_p, [_I32]),
    "fwa_device_count": (_I32, [ctypes.POINTER(_I32)]),
    "fwa_device_info": (_I32, [_I32, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_I32), ctypes.POINTER(_U64),
                               ctypes.POINTER(_I32)]),
    "fwa_ctx_create": (_I32, [_I32, _PP]),
    "fwa_ctx_destroy": (_I32, [_P]),
    "fwa_ctx_synchronize": (_I32, [_P]),
    "fwa_ctx_get_i64": (_I32, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "fwa_ctx_set_i64": (_I32, [_P, ctypes.c_char_p, ctypes.c_int64]),
    "fwa_ctx_peer_access": (_I32, [_P, _P, ctypes.POINTER(_I32)]),
    "fwa_ctx_device_info": (_I32, [_P, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_I32),
                                   ctypes.POINTER(_U64)]),
    "fwa_stream_create": (_I32, [_P, _PP]),
    "fwa_stream_wrap": (_I32, [_P, _P, _PP]),
    "fwa_stream_synchronize": (_I32, [_P]),
    "fwa_stream_destroy": (_I32, [_P]),
    "fwa_buf_alloc": (_I32, [_P, _U64, _PP]),
    "fwa_buf_wrap": (_I32, [_P, _P, _U64, _PP]),
    "fwa_buf_free": (_I32, [_P]),
    "fwa_buf_upload": (_I32, [_P, _U64, _P, _U64, _P]),
    "fwa_buf_download": (_I32, [_P, _P, _U64, _U64, _P]),
    "fwa_buf_copy": (_I32, [_P, _U64, _P, _U64, _U64, _P]),
    "fwa_host_alloc": (_I32, [_P, _U64, _PP]),
    "fwa_host_free": (_I32, [_P, _P]),
    "fwa_buf_download_async": (_I32, [_P, _P, _U64, _U64, _P]),
    "fwa_stream_wait_stream": (_I32, [_P, _P]),
    "fwa_buf_device_ptr": (_P, [_P]),
    "fwa_buf_size": (_U64, [_P]),
    "fwa_plan_create": (_I32, [_P, _I32, _U32, _P, _P, _PP]),
    "fwa_plan_exec": (_I32, [_P, _P, _PP]),
    "fwa_plan_destroy": (_I32, [_P]),
    "fwa_describe_path": (_I32, [_U32, ctypes.POINTER(_I32), ctypes.POINTER(_U32 * 3)]),
    "fwa_plan_get_i64": (_I32, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "fwa_plan_set_i64": (_I32, [_P, ctypes.c_char_p, ctypes.c_int64]),
    "fwa_slab": (_I32, [_U64, _I32, _I32, ctypes.POINTER(_U64), ctypes.POINTER(_U64)]),
    "fwa_comm_pieces": (_I32, [_U64, _U32, _I32, _I32, _I32, ctypes.POINTER(_U64), ctypes.POINTER(_U64), ctypes.POINTER(_I32),
                               ctypes.POINTER(_I32)]),
    "fwa_comm_unique_id": (_I32, [ctypes.c_char_p]),
    "fwa_comm_create": (_I32, [_P, ctypes.c_char_p, _I32, _I32, _PP]),
    "fwa_comm_destroy": (_I32, [_P]),
    "fwa_comm_get_i64": (_I32, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "fwa_comm_sendrecv": (_I32, [_P, _P, _U64, _U64, _I32, _P, _U64, _U64, _I32, _P]),
    "fwa_comm_scatter": (_I32, [_P, _I32, _P, _P, _U32, _U64, _P]),
    "fwa_comm_gather": (_I32, [_P, _I32, _P, _P, _U32, _U64, _P]),
    "fwa_event_create": (_I32, [_P, _PP]),
    "fwa_event_record": (_I32, [_P, _P]),
    "fwa_event_synchronize": (_I32, [_P]),
    "fwa_stream_wait_event": (_I32, [_P, _P]),
    "fwa_event_elapsed_ms": (_I32, [_P, _P, ctypes.POINTER(ctypes.c_float)]),
    "fwa_event_destroy": (_I32, [_P]),
    "fwa_fill_synthetic": (_I32, [_P, _U64, _U64, _U32, ctypes.c_float, _P]),
    "fwa_calib_copy": (_I32, [_P, _P, _U64, _P]),
}

_libs = {}


def lib(lab=False):
    """Load the shared library (once).  Raises if it was not built: no CPU fallback exists.  `lab`: False = the product,
    True = the laboratory build, a path = that build of the same sources (tools/build_variant.sh: A/B of compile-time defaults)."""
    path = lab if isinstance(lab, str) else (LAB_LIB_PATH if lab else LIB_PATH)
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C fft_wgpu_amd/csrc{' lab' if lab is True else ''}`.  fft_wgpu_amd has no CPU fallback.")
        L = ctypes.CDLL(path)
        # a stale library beside newer Python (or the reverse) is refused HERE, not at the first changed signature
        try:
            L.fwa_abi_version.restype = _I32
            L.fwa_abi_version.argtypes = []
            got = int(L.fwa_abi_version())
        except AttributeError:
            got = None
        if got != ABI_VERSION:
            raise RuntimeError(
                f"{path} reports ABI version {got}, this binding was written for version {ABI_VERSION} of "
                f"include/fft_wgpu_amd.h: rebuild the library (`make -C fft_wgpu_amd/csrc all lab`) from the same tree")
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(L, name)  # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        _libs[path] = L
    return _libs[path]


def check(status, ctx_handle, where, L=None):
    if status != FWA_OK:
        msg = (L or lib()).fwa_last_error_string(ctx_handle)
        raise FwaError(status, msg.decode() if msg else "", where)
