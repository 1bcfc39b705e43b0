"""ctypes binding of libxgcm_hip.so (the C ABI declared in include/xgcm_hip.h).

The reference has no native layer; its FFI for this path would be exactly this table.  cffi is
not available in the target image, so the stdlib `ctypes` is used.  There is deliberately NO
fallback: if the shared library is missing the import of any compute entry point raises.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
# XG_HIP_LIB points at another build of the same ABI (A/B measurements of kernel variants on one GPU)
LIB_PATH = os.environ.get("XG_HIP_LIB") or os.path.join(_HERE, "libxgcm_hip.so")

# enums of include/xgcm_hip.h
OP = {"diff": 0, "interp": 1, "min": 2, "max": 3, "minu": 4, "maxu": 5}  # minu / maxu: integer entry points, unsigned arrays
BC = {None: 0, "periodic": 1, "fill": 2, "extend": 3, "halo": 4}
BINOP = {"mul": 0, "div": 1, "add": 2, "sub": 3}
MAX_NDIM = 8

_i64p = C.POINTER(C.c_int64)
_intp = C.POINTER(C.c_int)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); the complete export list of include/xgcm_hip.h
SIGNATURES = {
    "xg_version": (C.c_int, []),
    "xg_last_error": (C.c_int, [C.c_char_p, C.c_int]),
    "xg_set_tunable": (C.c_int, [C.c_char_p, C.c_int]),
    "xg_get_tunable": (C.c_int, [C.c_char_p, _intp]),
    "xg_device_count": (C.c_int, []),
    "xg_set_device": (C.c_int, [C.c_int]),
    "xg_malloc": (C.c_int, [C.POINTER(_vp), C.c_uint64]),
    "xg_free": (C.c_int, [_vp]),
    "xg_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "xg_pin_host": (C.c_int, [_vp, C.c_uint64]),
    "xg_unpin_host": (C.c_int, [_vp]),
    "xg_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "xg_stream_sync": (C.c_int, [_vp]),
    "xg_scatter_alloc": (C.c_int, [C.POINTER(_vp), C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]),
    "xg_scatter_free": (C.c_int, [_vp]),
    "xg_scatter_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xg_scatter_grade": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]),
    "xg_scatter_grade_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xg_pool_alloc": (_vp, [C.c_ssize_t, C.c_int, _vp]),
    "xg_pool_free": (None, [_vp, C.c_ssize_t, C.c_int, _vp]),
    "xg_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "xg_stream_destroy": (C.c_int, [_vp]),
    "xg_chain_status": (C.c_int, [_intp, _intp]),
    "xg_chain_rearm": (C.c_int, []),
    "xg_event_create": (C.c_int, [C.POINTER(_vp)]),
    "xg_event_record": (C.c_int, [_vp, _vp]),
    "xg_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "xg_event_destroy": (C.c_int, [_vp]),
    "xg_bswap": (C.c_int, [_vp, C.c_uint64, C.c_int, _vp]),
    "xg_mask_value": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_double, _vp]),
    "xg_stencil1d_f64": (
        C.c_int,
        [C.c_int, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double,
         _vp, _i64p, _vp, _i64p, _vp],
    ),
    "xg_stencil1d_halo_f64": (
        C.c_int,
        [C.c_int, _vp, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _vp, _i64p, _vp],
    ),
    "xg_stencil1d_halo_w_f64": (
        C.c_int,
        [C.c_int, _vp, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _vp, _i64p, _vp, _i64p, _vp],
    ),
    "xg_cumsum1d_f64": (
        C.c_int,
        [_vp, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_double, _vp, _i64p, _vp, _i64p, _vp],
    ),
    "xg_reduce1d_f64": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_int, C.c_int, _vp, _i64p, _vp]),
    "xg_pad_f64": (C.c_int, [_vp, _vp, _i64p, C.c_int, _i64p, _i64p, _intp, _f64p, _intp, _vp]),
    "xg_gather_f64": (
        C.c_int,
        [_vp, _vp, _vp, _i64p, _i64p, _i64p, C.c_int, _intp, _intp, _i64p, _vp, C.c_int64, _f64p, C.c_int, _vp],
    ),
    "xg_halo_put_f64": (C.c_int, [_vp, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "xg_transform_linear_f64": (
        C.c_int,
        [_vp, _vp, _i64p, _vp, _i64p, C.c_int64, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    ),
    "xg_transform_conservative_f64": (
        C.c_int, [_vp, _vp, _i64p, _vp, C.c_int64, _vp, _i64p, C.c_int, C.c_int, _vp],
    ),
    "xg_binary_f64": (C.c_int, [C.c_int, _vp, _i64p, _vp, _i64p, _vp, _i64p, C.c_int, _vp]),
    "xg_vorticity_f64": (
        C.c_int,
        [_vp, _vp, _vp, _i64p, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp],
    ),
    "xg_divergence_f64": (
        C.c_int,
        [_vp, _vp, _vp, _i64p, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp],
    ),
    "xg_vorticity_halo_f64": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp],
    ),
    "xg_divergence_halo_f64": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp],
    ),
    "xg_gradient_f64": (
        C.c_int,
        [_vp, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp, _i64p, _vp, _i64p, _vp],
    ),
    "xg_flux_f64": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp],
    ),
    "xg_gradient_halo_f64": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp, _i64p, _vp, _i64p, _vp],
    ),
    "xg_flux_halo_f64": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _vp],
    ),
    "xg_stencil2d_f64": (
        C.c_int,
        [C.c_int, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
         C.c_int, C.c_double, _vp],
    ),
    "xg_stencil2d_metric_f64": (
        C.c_int,
        [C.c_int, _vp, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
         C.c_int, C.c_double, _vp, _vp, _vp, _vp],
    ),
    "xg_fill_synthetic_f64": (C.c_int, [_vp, C.c_int64, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _vp]),
}

# float32 twins of the compute entry points: same argument order, `float` fill values
# (xg_fill_synthetic_f32 keeps double scale/shift: the value is formed in f64 and rounded once)
for _name in ["xg_stencil1d", "xg_stencil1d_halo", "xg_stencil1d_halo_w", "xg_cumsum1d", "xg_reduce1d", "xg_pad", "xg_gather", "xg_halo_put", "xg_transform_linear", "xg_transform_conservative", "xg_binary", "xg_vorticity", "xg_divergence", "xg_vorticity_halo", "xg_divergence_halo", "xg_gradient", "xg_flux", "xg_gradient_halo", "xg_flux_halo", "xg_stencil2d", "xg_stencil2d_metric",
              "xg_fill_synthetic"]:
    _res, _args = SIGNATURES[_name + "_f64"]
    SIGNATURES[_name + "_f32"] = (_res, list(_args) if _name == "xg_fill_synthetic" else
                                  [C.c_float if a is C.c_double else (C.POINTER(C.c_float) if a is _f64p else a) for a in _args])

# int64 twins of the entry points that serve integer arrays (numpy keeps them integral: diff / min / max / cumsum / pad /
# gather / +,-,*): same argument order, int64 fill values; metric / weight pointers must be NULL (include/xgcm_hip.h)
_i64p_fill = C.POINTER(C.c_int64)
for _name in ["xg_stencil1d", "xg_stencil1d_halo", "xg_cumsum1d", "xg_reduce1d", "xg_pad", "xg_gather", "xg_halo_put", "xg_binary"]:
    _res, _args = SIGNATURES[_name + "_f64"]
    SIGNATURES[_name + "_i64"] = (_res, [C.c_int64 if a is C.c_double else (_i64p_fill if a is _f64p else a) for a in _args])
# int32 twins: the entry points whose integer result keeps the array's width (scans / sums accumulate in 64 bits)
_i32p_fill = C.POINTER(C.c_int32)
for _name in ["xg_stencil1d", "xg_stencil1d_halo", "xg_pad", "xg_gather", "xg_halo_put", "xg_binary"]:
    _res, _args = SIGNATURES[_name + "_f64"]
    SIGNATURES[_name + "_i32"] = (_res, [C.c_int32 if a is C.c_double else (_i32p_fill if a is _f64p else a) for a in _args])

# element types of xg_convert (enum xg_dtype), keyed by numpy dtype name
DTYPE = {"bool": 0, "int8": 1, "int16": 2, "int32": 3, "int64": 4, "uint8": 5, "uint16": 6, "uint32": 7, "uint64": 8,
         "float32": 9, "float64": 10, "float16": 11}
SIGNATURES["xg_copy_nd"] = (C.c_int, [_vp, _i64p, _vp, _i64p, _i64p, C.c_int, C.c_int, _vp])
SIGNATURES["xg_convert"] = (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int, _vp])

SUFFIX = {"float64": "f64", "float32": "f32", "int64": "i64", "int32": "i32"}

_lib: Optional[C.CDLL] = None


class XgcmHipError(RuntimeError):
    """A C-ABI call returned a negative status (message from xg_last_error)."""


class XgcmInvalidArgument(XgcmHipError, ValueError):
    """XG_ERR_INVALID: the library refused its arguments (an empty axis padded with 'wrap' / 'edge', widths that do not
    fit ...).  numpy raises ValueError for the same requests, and code written against the reference catches that."""


def error_for(status: int, text: str) -> XgcmHipError:
    return (XgcmInvalidArgument if status == -1 else XgcmHipError)(text)


def load() -> C.CDLL:
    """Load the shared library (once) and type every export.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "xgcm_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.xg_version() != 1:
        raise ImportError(f"libxgcm_hip.so ABI version {lib.xg_version()} != 1")
    _lib = lib
    return lib


def set_tunable(name: str, value: int) -> None:
    """launch-shape tunable of the library (speed only, never results); see INTEGRATION.md"""
    check(load().xg_set_tunable(name.encode(), int(value)))


def get_tunable(name: str) -> int:
    v = C.c_int(0)
    check(load().xg_get_tunable(name.encode(), C.byref(v)))
    return int(v.value)


class ChainRescueWarning(RuntimeWarning):
    """A chained scan / reduction gave up on a hand-off and was redone, in stream, by its marching twin."""


_chain_reported = [0, 0]


def chain_status():
    """(gave_up, redone) of xg_chain_status: sticky "some chunk of a chained launch gave up", launches redone so far"""
    g, r = C.c_int(0), C.c_int(0)
    check(load().xg_chain_status(C.byref(g), C.byref(r)))
    return int(g.value), int(r.value)


def chain_check() -> None:
    """Called wherever results leave the library towards the host (tohost / .values, graph replays, stream syncs of
    the streaming paths): warn ONCE per event that a chained launch had to be redone.  The data the caller reads is
    the marching kernel's -- correct -- so this is a warning, not an error (include/xgcm_hip.h: xg_chain_status)."""
    if _lib is None:
        return
    g, r = chain_status()
    if g > _chain_reported[0] or r > _chain_reported[1]:
        _chain_reported[0], _chain_reported[1] = g, r
        import warnings

        warnings.warn(
            f"xgcm_amd: a chained scan / reduction kernel gave up waiting for a predecessor chunk; {r} launch(es) so far "
            "were redone by the marching kernel on the same stream (results are correct).  The library plans marching "
            "kernels from now on; xgcm_amd._hip.chain_rearm() re-enables the chained ones, XG_SCAN_CHAIN=0 avoids them.",
            ChainRescueWarning, stacklevel=3)


def chain_rearm() -> None:
    check(load().xg_chain_rearm())
    _chain_reported[0] = 0


def scatter_stats() -> dict:
    """scattered result buffers made so far, bytes alive, and pool requests that fell back to plain hipMalloc"""
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    check(load().xg_scatter_stats(C.byref(a), C.byref(b), C.byref(c)))
    g, r = C.c_uint64(0), C.c_uint64(0)
    check(load().xg_scatter_grade_stats(C.byref(g), C.byref(r)))
    # graded: pool requests of 1 GiB or more whose buffer was graded; rejected: buffers parked and released on the way to a good one
    return {"buffers_made": int(a.value), "live_bytes": int(b.value), "pool_fallbacks": int(c.value), "graded": int(g.value),
            "rejected": int(r.value)}


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().xg_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(status: int) -> None:
    if status != 0:
        raise error_for(status, f"xgcm_hip status {status}: {last_error()}")


def i64(values: Optional[Sequence[int]]):
    if values is None:
        return None
    return (C.c_int64 * len(values))(*[int(v) for v in values])


def ints(values: Optional[Sequence[int]]):
    if values is None:
        return None
    return (C.c_int * len(values))(*[int(v) for v in values])


def reals(values: Optional[Sequence[float]], suffix: str):
    """array of fill values in the C type of the `_f64` / `_f32` entry point"""
    if values is None:
        return None
    if suffix == "i64":
        return (C.c_int64 * len(values))(*[int(v) for v in values])
    if suffix == "i32":
        return (C.c_int32 * len(values))(*[int(v) for v in values])
    ctype = C.c_double if suffix == "f64" else C.c_float
    return (ctype * len(values))(*[float(v) for v in values])
