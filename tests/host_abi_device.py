"""Test-side stand-in for `xgcm_amd.device` that calls the HOST build of the C ABI (xgcm_amd/libxgcm_host.so:
the symbols of include/xgcm_hip.h over host pointers, compiled by g++ from xgcm_amd/csrc/xg_host.cpp).

TEST INFRASTRUCTURE ONLY: it lets the CPU suite run the whole `Grid` stack -- dispatch, signatures, argument
marshalling, strides, error paths -- through a real shared library with the real binding table
(`xgcm_amd._hip.SIGNATURES`) instead of the numpy oracle (BASELINE config 1: "plumbing, no GPU").  The product
never loads libxgcm_host.so; on a GPU box the same tests run against libxgcm_hip.so."""

import ctypes as C
import os

import numpy as np

from xgcm_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "xgcm_amd", "libxgcm_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _hip.SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _check(rc):
    if rc != 0:
        buf = C.create_string_buffer(512)
        lib().xg_last_error(buf, 512)
        raise _hip.XgcmHipError(f"xgcm host ABI status {rc}: {buf.value.decode(errors='replace')}")


def _common(*arrays):
    present = [np.asarray(a) for a in arrays if a is not None]
    f32 = bool(present) and all(a.dtype == np.float32 for a in present)
    return (np.float32, "f32") if f32 else (np.float64, "f64")


def asdevice(x, dtype=None):
    a = np.asarray(x)
    if dtype is None:
        dtype = np.float32 if a.dtype == np.float32 else np.float64
    return np.ascontiguousarray(a, dtype=dtype)


def tohost(x):
    return np.asarray(x)


def is_device_array(x):
    return False


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _strides(m, shape, what):
    """element strides of a dim-aligned metric against `shape` (0 = broadcast), like device._bstrides"""
    if m is None:
        return None
    if m.ndim != len(shape):
        raise ValueError(f"{what}: metric has {m.ndim} dims, array has {len(shape)}")
    st = []
    for d, (ms, s) in enumerate(zip(m.shape, shape)):
        if ms == s and s != 1:
            st.append(m.strides[d] // m.itemsize)
        elif ms == 1:
            st.append(0)
        else:
            raise ValueError(f"{what}: metric extent {ms} does not broadcast against {s} on dim {d}")
    return st


def stencil1d(op, x, axis, pad_lo, pad_hi, bc, fill=0.0, m_in=None, m_out=None):
    dt, sfx = _common(x, m_in, m_out)
    x = asdevice(x, dt)
    axis %= x.ndim
    shape = list(x.shape)
    oshape = list(shape)
    oshape[axis] = shape[axis] + pad_lo + pad_hi - 1
    m_in = None if m_in is None else asdevice(m_in, dt)
    m_out = None if m_out is None else asdevice(m_out, dt)
    out = np.empty(oshape, dtype=dt)
    if out.size == 0:
        return out
    _check(getattr(lib(), "xg_stencil1d_" + sfx)(
        _hip.OP[op], _ptr(x), _ptr(out), _hip.i64(shape), len(shape), axis, oshape[axis], int(pad_lo), int(pad_hi),
        _hip.BC[bc], float(fill), _ptr(m_in), _hip.i64(_strides(m_in, shape, "m_in")), _ptr(m_out),
        _hip.i64(_strides(m_out, oshape, "m_out")), None))
    return out


def cumsum1d(x, axis, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill=0.0, reverse=False, skipna=True, m_in=None, m_out=None):
    dt, sfx = _common(x, m_in, m_out)
    x = asdevice(x, dt)
    axis %= x.ndim
    shape = list(x.shape)
    oshape = list(shape)
    oshape[axis] = shape[axis] - trim_lo - trim_hi + pad_lo + pad_hi
    m_in = None if m_in is None else asdevice(m_in, dt)
    m_out = None if m_out is None else asdevice(m_out, dt)
    out = np.empty(oshape, dtype=dt)
    if out.size == 0:
        return out
    _check(getattr(lib(), "xg_cumsum1d_" + sfx)(
        _ptr(x), _ptr(out), _hip.i64(shape), len(shape), axis, int(bool(reverse)), int(bool(skipna)), int(trim_lo),
        int(trim_hi), int(pad_lo), int(pad_hi), _hip.BC[bc], float(fill), _ptr(m_in),
        _hip.i64(_strides(m_in, shape, "m_in")), _ptr(m_out), _hip.i64(_strides(m_out, oshape, "m_out")), None))
    return out


def reduce1d(x, axis, w=None, skipna=True):
    dt, sfx = _common(x, w)
    x = asdevice(x, dt)
    axis %= x.ndim
    shape = list(x.shape)
    w = None if w is None else asdevice(w, dt)
    mode = {"valid": 2, "all": 3, "mean_valid": 4, "mean_all": 5, "pair_valid": 6, "pair_all": 7}.get(skipna, int(bool(skipna)))
    out = np.empty(([2] if mode >= 6 else []) + shape[:axis] + shape[axis + 1:], dtype=dt)
    if out.size == 0:
        return out
    _check(getattr(lib(), "xg_reduce1d_" + sfx)(_ptr(x), _ptr(out), _hip.i64(shape), len(shape), axis, mode, _ptr(w),
                                               _hip.i64(_strides(w, shape, "w")), None))
    return out


def binary(op, a, b):
    dt, sfx = _common(a, b)
    a, b = asdevice(a, dt), asdevice(b, dt)
    shape = [max(sa, sb) if 0 not in (sa, sb) else 0 for sa, sb in zip(a.shape, b.shape)]
    out = np.empty(shape, dtype=dt)
    if out.size == 0:
        return out
    _check(getattr(lib(), "xg_binary_" + sfx)(_hip.BINOP[op], _ptr(a), _hip.i64(_strides(a, shape, "a")), _ptr(b),
                                             _hip.i64(_strides(b, shape, "b")), _ptr(out), _hip.i64(shape), len(shape), None))
    return out


def pad_nd(x, widths, bc, fill):
    dt, sfx = _common(x)
    x = asdevice(x, dt)
    nd = x.ndim
    lo, hi, bcv, fv, order = [0] * nd, [0] * nd, [0] * nd, [0.0] * nd, []
    for ax, (l, h) in widths.items():
        ax %= nd
        lo[ax], hi[ax] = int(l), int(h)
        bcv[ax] = _hip.BC[bc.get(ax)]
        f = fill.get(ax, 0.0)
        fv[ax] = 0.0 if f is None else float(f)
        order.append(ax)
    order += [d for d in range(nd) if d not in order]
    out = np.empty([s + l + h for s, l, h in zip(x.shape, lo, hi)], dtype=dt)
    if out.size == 0:
        return out
    _check(getattr(lib(), "xg_pad_" + sfx)(_ptr(x), _ptr(out), _hip.i64(list(x.shape)), nd, _hip.i64(lo), _hip.i64(hi),
                                          _hip.ints(bcv), _hip.reals(fv, sfx), _hip.ints(order), None))
    return out


def synthetic(shape, seed, offset=0, scale=1.0, shift=-0.5, out=None, dtype=np.float64):
    out = np.empty(tuple(shape), dtype=dtype) if out is None else out
    sfx = "f32" if out.dtype == np.float32 else "f64"
    _check(getattr(lib(), "xg_fill_synthetic_" + sfx)(_ptr(out), out.size, int(seed), int(offset), float(scale), float(shift), None))
    return out


def stencil2d_supported(x, padx, pady):
    return False  # the host build has no fused two-axis entry point: the Grid runs the axes one after the other


def _not_in_host_build(name):
    def f(*a, **k):
        raise _hip.XgcmHipError(f"{name}: not part of the host build of the ABI")
    return f


_SERVED = ["asdevice", "tohost", "is_device_array", "stencil1d", "cumsum1d", "reduce1d", "pad_nd", "binary", "synthetic",
           "stencil2d_supported"]
_ABSENT = ["stencil1d_halo", "gather", "upload_tokens", "transform_linear", "transform_conservative", "vorticity",
           "divergence", "gradient", "flux", "stencil2d"]


def install(monkeypatch):
    import xgcm_amd.device as dev

    for n in _SERVED:
        monkeypatch.setattr(dev, n, globals()[n])
    for n in _ABSENT:
        monkeypatch.setattr(dev, n, _not_in_host_build(n))
