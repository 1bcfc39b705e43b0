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
from xgcm_amd import dtypes as _dt

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
        raise _hip.error_for(rc, f"xgcm host ABI status {rc}: {buf.value.decode(errors='replace')}")


def _in(x):
    """the product's device._raw_device over host memory: a contiguous array in NATIVE byte order; an array of the other
    byte order is copied as raw bytes and reversed by the host build of xg_bswap -- the SAME intake rule
    (xgcm_amd.dtypes.host_intake), never a reinterpretation by dtype name"""
    a, swap = _dt.host_intake(np.asarray(x, order="C"))
    if swap:
        a = a.copy()
        if a.size:
            _check(lib().xg_bswap(_ptr(a), a.size, swap, None))
    return a


def _common(*arrays):
    present = [_dt.np_dtype(a) for a in arrays if a is not None]
    f = _dt.float_of(*present)
    return (np.float32, "f32") if f == np.float32 else (np.float64, "f64")


def _is_int(x):
    return x is not None and _dt.is_integer(_dt.np_dtype(x))


def _half(*arrays):
    return _dt.half_result(*[_dt.np_dtype(a) for a in arrays if a is not None])


def _out(res, half):
    return convert(res, np.float16) if half else res


def _steps(x, m_in, m_out):
    return _dt.metric_steps(_dt.np_dtype(x), None if m_in is None else _dt.np_dtype(m_in),
                            None if m_out is None else _dt.np_dtype(m_out))


def convert(x, dst, via=None, scale=1.0, flip=False):
    """numpy `astype` through the host build of xg_convert (the product's device.convert over host pointers)"""
    a = _in(x)
    dst = np.dtype(dst)
    if a.dtype == dst and via is None and scale == 1.0 and not flip:
        return a
    out = np.empty(a.shape, dtype=dst)
    if out.size:
        _check(lib().xg_convert(_ptr(a), _hip.DTYPE[a.dtype.name], _ptr(out), _hip.DTYPE[dst.name], a.size,
                                -1 if via is None else _hip.DTYPE[np.dtype(via).name], float(scale), 1 if flip else 0, None))
    return out


def asdevice(x, dtype=None):
    a = _in(x)
    if dtype is None:
        return a
    return a if a.dtype == dtype else convert(a, dtype)


_LANE_SFX = {"int64": "i64", "int32": "i32"}


def _widen(x, lane=np.int64):
    a = _in(x)
    lane = np.dtype(lane)
    if _dt.same_bits(a.dtype, lane):
        return a.view(lane)
    return convert(a, lane)


def _narrow(t, dst, via=None, scale=1.0):
    dst = np.dtype(dst)
    if via is None and scale == 1.0 and _dt.same_bits(dst, t.dtype):
        return t.view(dst)
    return convert(t, dst, via=via, scale=scale)


def _lane_int(value, lane=np.int64):
    bits = 8 * np.dtype(lane).itemsize
    v = int(value) & ((1 << bits) - 1)
    return v - (1 << bits) if v >= (1 << (bits - 1)) else v


def _divide(res, m_out, as_dtype):
    return res if m_out is None else binary("div", convert(res, as_dtype), m_out)


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
    plan = _dt.stencil_plan(op, _dt.np_dtype(x), None if m_in is None else _dt.np_dtype(m_in),
                            None if m_out is None else _dt.np_dtype(m_out))
    if plan.lanes == "int":  # the product's device._int_stencil1d over the host build of the *_i64 entry points
        src = _dt.np_dtype(x)
        lane = plan.compute
        t = _widen(x, lane)
        axis %= t.ndim
        shape = list(t.shape)
        oshape = list(shape)
        oshape[axis] = shape[axis] + pad_lo + pad_hi - 1
        out = np.empty(oshape, dtype=lane)
        if out.size:
            fv = _lane_int(_dt.fill_as(src, fill), lane) if (bc == "fill" and (pad_lo or pad_hi)) else 0
            code = _hip.OP[op + "u"] if plan.unsigned else _hip.OP[op]
            _check(getattr(lib(), "xg_stencil1d_" + _LANE_SFX[lane.name])(code, _ptr(t), _ptr(out), _hip.i64(shape), len(shape), axis, oshape[axis],
                                          int(pad_lo), int(pad_hi), _hip.BC[bc], fv, None, None, None, None, None))
        return _divide(_narrow(out, plan.result, via=plan.via, scale=plan.scale), m_out, plan.divide_as)
    pre_mul, post_div = _steps(x, m_in, m_out)
    if pre_mul:
        x, m_in = binary("mul", x, m_in), None
    if post_div:
        return binary("div", stencil1d(op, x, axis, pad_lo, pad_hi, bc, fill, m_in, None), m_out)
    half = _half(x, m_in, m_out)
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
        return _out(out, half)
    _check(getattr(lib(), "xg_stencil1d_" + sfx)(
        _hip.OP[op], _ptr(x), _ptr(out), _hip.i64(shape), len(shape), axis, oshape[axis], int(pad_lo), int(pad_hi),
        _hip.BC[bc], float(fill), _ptr(m_in), _hip.i64(_strides(m_in, shape, "m_in")), _ptr(m_out),
        _hip.i64(_strides(m_out, oshape, "m_out")), None))
    return _out(out, half)


def cumsum1d(x, axis, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill=0.0, reverse=False, skipna=True, m_in=None, m_out=None):
    if _is_int(x) and m_in is None:
        res_dt = _dt.cumsum_dtype(_dt.np_dtype(x))
        t = _widen(x)
        axis %= t.ndim
        shape = list(t.shape)
        oshape = list(shape)
        oshape[axis] = shape[axis] - trim_lo - trim_hi + pad_lo + pad_hi
        fv = _lane_int(_dt.fill_as(res_dt, fill)) if (bc == "fill" and (pad_lo or pad_hi)) else 0
        out = np.empty(oshape, dtype=np.int64)
        if out.size:
            _check(lib().xg_cumsum1d_i64(_ptr(t), _ptr(out), _hip.i64(shape), len(shape), axis, int(bool(reverse)), 0,
                                         int(trim_lo), int(trim_hi), int(pad_lo), int(pad_hi), _hip.BC[bc], fv, None, None,
                                         None, None, None))
        return _divide(_narrow(out, res_dt), m_out, None if m_out is None else _dt.float_of(res_dt, _dt.np_dtype(m_out)))
    pre_mul, post_div = _steps(x, m_in, m_out)
    if pre_mul:
        x, m_in = binary("mul", x, m_in), None
    if post_div:
        return binary("div", cumsum1d(x, axis, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill, reverse, skipna, m_in, None), m_out)
    half = _half(x, m_in, m_out)
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
        return _out(out, half)
    _check(getattr(lib(), "xg_cumsum1d_" + sfx)(
        _ptr(x), _ptr(out), _hip.i64(shape), len(shape), axis, int(bool(reverse)), int(bool(skipna)), int(trim_lo),
        int(trim_hi), int(pad_lo), int(pad_hi), _hip.BC[bc], float(fill), _ptr(m_in),
        _hip.i64(_strides(m_in, shape, "m_in")), _ptr(m_out), _hip.i64(_strides(m_out, oshape, "m_out")), None))
    return _out(out, half)


def reduce1d(x, axis, w=None, skipna=True):
    if _is_int(x) and w is None and isinstance(skipna, (bool, int, np.bool_)):
        res_dt = _dt.cumsum_dtype(_dt.np_dtype(x))
        t = _widen(x)
        axis %= t.ndim
        shape = list(t.shape)
        out = np.zeros(shape[:axis] + shape[axis + 1:], dtype=np.int64)
        if out.size and t.size:
            _check(lib().xg_reduce1d_i64(_ptr(t), _ptr(out), _hip.i64(shape), len(shape), axis, 0, None, None, None))
        return _narrow(out, res_dt)
    half = _half(x, w)
    if half and w is not None and skipna not in ("valid", "all"):
        x, w = binary("mul", x, w), None
    dt, sfx = _common(x, w)
    x = asdevice(x, dt)
    axis %= x.ndim
    shape = list(x.shape)
    w = None if w is None else asdevice(w, dt)
    mode = {"valid": 2, "all": 3, "mean_valid": 4, "mean_all": 5, "pair_valid": 6, "pair_all": 7}.get(skipna, int(bool(skipna)))
    out = np.empty(([2] if mode >= 6 else []) + shape[:axis] + shape[axis + 1:], dtype=dt)
    if out.size == 0:
        return _out(out, half)
    _check(getattr(lib(), "xg_reduce1d_" + sfx)(_ptr(x), _ptr(out), _hip.i64(shape), len(shape), axis, mode, _ptr(w),
                                               _hip.i64(_strides(w, shape, "w")), None))
    return _out(out, half)


def binary(op, a, b):
    lanes, res_dt = _dt.binary_plan(op, _dt.np_dtype(a), _dt.np_dtype(b))
    half = False
    if lanes == "int":
        lane = _dt.lane_of(res_dt)
        dt, sfx = lane, _LANE_SFX[lane.name]
        a = _widen(a if _dt.same_bits(_dt.np_dtype(a), lane) else convert(a, res_dt), lane)
        b = _widen(b if _dt.same_bits(_dt.np_dtype(b), lane) else convert(b, res_dt), lane)
    else:
        dt, sfx = (np.float32, "f32") if res_dt == np.float32 else (np.float64, "f64")
        half = _half(a, b)
        a, b = asdevice(a, dt), asdevice(b, dt)
    if a.ndim == 0 and b.ndim == 0:  # (mirrors device.binary: two scalars are one cell of a 1-d launch)
        return binary(op, a.reshape(1), b.reshape(1)).reshape(())
    shape = [max(sa, sb) if 0 not in (sa, sb) else 0 for sa, sb in zip(a.shape, b.shape)]
    out = np.empty(shape, dtype=dt)
    if out.size:
        _check(getattr(lib(), "xg_binary_" + sfx)(_hip.BINOP[op], _ptr(a), _hip.i64(_strides(a, shape, "a")), _ptr(b),
                                                 _hip.i64(_strides(b, shape, "b")), _ptr(out), _hip.i64(shape), len(shape), None))
    return _narrow(out, res_dt) if lanes == "int" else _out(out, half)


def pad_nd(x, widths, bc, fill):
    src = _dt.np_dtype(x)
    ints = _dt.is_integer(src)
    if ints:
        lane = _dt.lane_of(src)
        dt, sfx, x = lane, _LANE_SFX[lane.name], _widen(x, lane)
    else:
        dt, sfx = _common(x)
        x = asdevice(x, dt)
    nd = x.ndim
    lo, hi, bcv, fv, order = [0] * nd, [0] * nd, [0] * nd, [0 if ints else 0.0] * nd, []
    for ax, (l, h) in widths.items():
        ax %= nd
        lo[ax], hi[ax] = int(l), int(h)
        bcv[ax] = _hip.BC[bc.get(ax)]
        f = fill.get(ax, 0.0)
        f = 0.0 if f is None else f
        fv[ax] = _lane_int(_dt.fill_as(src, f), lane) if ints else float(f)
        order.append(ax)
    order += [d for d in range(nd) if d not in order]
    out = np.empty([s + l + h for s, l, h in zip(x.shape, lo, hi)], dtype=dt)
    if out.size:
        _check(getattr(lib(), "xg_pad_" + sfx)(_ptr(x), _ptr(out), _hip.i64(list(x.shape)), nd, _hip.i64(lo), _hip.i64(hi),
                                              _hip.ints(bcv), _hip.reals(fv, sfx), _hip.ints(order), None))
    return _narrow(out, src) if ints else _out(out, src == np.float16)


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
_ABSENT = ["stencil1d_halo", "gather", "put_halo", "upload_tokens", "transform_linear", "transform_conservative", "vorticity",
           "divergence", "gradient", "flux", "stencil2d"]


def install(monkeypatch):
    import xgcm_amd.device as dev

    for n in _SERVED:
        monkeypatch.setattr(dev, n, globals()[n])
    for n in _ABSENT:
        monkeypatch.setattr(dev, n, _not_in_host_build(n))
