"""dtype policy of the operators: which lanes an array computes on and what dtype numpy would return.

The reference runs its bodies in the array's own dtype (xgcm/gridops.py:23-24,76-77,123-126,172-175,227-278 are plain
numpy expressions; `np.pad` keeps the dtype and casts the fill value, xgcm/padding.py:610-615), so integer and bool arrays
stay integral through diff / min / max / cumsum / pad -- wrapping modulo 2^bits -- and become float64 in `interp`
(`/ 2.0`) or next to a metric (`int * float64`).  The kernels compute on float64, float32, int64 or int32 lanes; this
module decides, from dtypes alone, which lanes serve a call and how the result leaves them (pure host logic: no device,
no arithmetic on array data).  `xgcm_amd.device` executes the plans with xg_convert / the `*_i64` / `*_i32` entry points.
"""

from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np

try:  # torch is plumbing for device memory only
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore

INT64 = np.dtype(np.int64)
INT32 = np.dtype(np.int32)
UINT64 = np.dtype(np.uint64)
FLOAT64 = np.dtype(np.float64)
FLOAT32 = np.dtype(np.float32)
FLOAT16 = np.dtype(np.float16)
BOOL = np.dtype(np.bool_)

_TORCH_TO_NUMPY = {}
if torch is not None:
    for _n in ("bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float16", "float32", "float64"):
        if hasattr(torch, _n):
            _TORCH_TO_NUMPY[getattr(torch, _n)] = np.dtype(_n)
    _TORCH_TO_NUMPY[torch.bfloat16] = np.dtype(np.float32)  # no numpy twin: treated as its float32 promotion


def native(dt) -> np.dtype:
    """`dt` in the machine's byte order: what every numpy operation RETURNS for an array of dtype `dt` (the reference's
    bodies are numpy expressions, xgcm/gridops.py:23-24,76-77: a big-endian field in, a native result out)"""
    dt = np.dtype(dt)
    return dt if dt.isnative else dt.newbyteorder("=")


def np_dtype(x) -> np.dtype:
    """numpy dtype, in NATIVE byte order, of a numpy array, a torch tensor, a python scalar or a dtype: everything that
    plans lanes and result dtypes reasons about values, and `>f8` holds the values of float64"""
    if torch is not None and isinstance(x, torch.Tensor):
        return _TORCH_TO_NUMPY[x.dtype]
    if isinstance(x, np.dtype):
        return native(x)
    dt = getattr(x, "dtype", None)
    if dt is not None and isinstance(dt, np.dtype):
        return native(dt)
    return native(np.asarray(x).dtype)


SERVED = ("bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float16", "float32", "float64")


def check_served(dt, what: str = "array") -> np.dtype:
    """native twin of `dt` if the kernels serve it, else a TypeError that names the dtype (complex, object, str,
    datetime64, float128 ...: numpy would compute some of them, this backend does not -- say so instead of failing on a
    table lookup further down)"""
    dt = native(dt)
    if dt.name not in SERVED:
        kind = "complex arrays are" if dt.kind == "c" else f"dtype {dt} is"
        raise TypeError(f"{what}: {kind} not supported by the MI355X backend (served: bool, (u)int8-64, float16 / 32 / 64, "
                        "in either byte order)")
    return dt


def host_intake(a: np.ndarray):
    """(`a`'s bytes seen as a NATIVE-order array -- a view, no copy --, element size if the byte order must be reversed
    after the copy else 0).  ONE intake rule for every backend (HIP: PCIe copy of the raw bytes, `xg_bswap` in HBM; host
    ABI double: copy, `xg_bswap` of the host build): an array of non-native byte order -- `np.fromfile(f, ">f4")`, an
    MDS / NetCDF-3 record, xmitgcm output -- is never reinterpreted by its dtype NAME (`np.dtype(">f8").name` is
    "float64")."""
    dt = check_served(a.dtype)
    if a.dtype.isnative or a.dtype.itemsize == 1:
        return (a if a.dtype == dt else a.view(dt)), 0
    return a.view(dt), a.dtype.itemsize


def torch_dtype(dt):
    return getattr(torch, np.dtype(dt).name)


def is_integer(dt) -> bool:
    """bool, signed or unsigned integer: the dtypes numpy keeps integral"""
    return np.dtype(dt).kind in "biu"


def float_of(*dtypes) -> np.dtype:
    """the float lanes a mix of operands computes on: numpy's array-array promotion, float32 only when it yields
    float32 (all-float32 operands, or float32 next to 8 / 16-bit integers) or float16 (no float16 lanes exist: float16
    arrays are widened to float32, see `half_result`), float64 for everything else"""
    rt = np.result_type(*[native(d) for d in dtypes]) if dtypes else FLOAT64
    return FLOAT32 if rt in (FLOAT32, FLOAT16) else FLOAT64


def half_result(*dtypes) -> bool:
    """does numpy return float16 for this mix of operands (all float16, or float16 next to bool / 8-bit integers)?
    Then the float32-lane result is narrowed to float16 on the way out: every single numpy operation on float16 is the
    float32 operation rounded once (numpy's half loops do exactly that), so `diff`, `min`, `max`, `pad`, `a OP b` are bit
    for bit numpy's; `interp` differs where `a + b` overflows float16 (numpy: inf, here the finite mean), and sums /
    prefix sums carry float32 partial sums where numpy rounds every partial sum to float16 (stated tolerance in
    tests/test_dtype_intake.py)."""
    present = [native(d) for d in dtypes if d is not None]
    return bool(present) and np.result_type(*present) == FLOAT16


def metric_steps(x_dt, m_in_dt=None, m_out_dt=None):
    """(pre_mul, post_div): numpy evaluates `(x * m_in)` -> body -> `/ m_out` (xgcm/grid.py:804-808,830-832) one operation
    at a time, each in ITS promoted dtype and rounded to it.  The kernels fuse the three on ONE set of lanes, which is the
    same arithmetic only when every step runs in that dtype; otherwise the step numpy rounds differently runs on its own:
    `x * m_in` first when the product is float16, `/ m_out` afterwards when the quotient is wider than the body's dtype
    (float32 field, float64 metric: the difference is rounded to float32 BEFORE it is divided) or float16."""
    x_dt = native(x_dt)
    pre = x_dt if m_in_dt is None else np.result_type(x_dt, native(m_in_dt))
    pre_mul = m_in_dt is not None and pre == FLOAT16
    post_div = m_out_dt is not None and (pre == FLOAT16 or float_of(pre, m_out_dt) != float_of(pre))
    return pre_mul, post_div


def fill_as(dt, fill):
    """the constant numpy.pad writes into an array of dtype `dt` for `constant_values=fill` (xgcm/padding.py:610-615):
    truncation toward zero for integers, OverflowError / ValueError for values the dtype cannot hold -- obtained from
    numpy.pad itself on a one-cell array, so every corner (bool, negative into unsigned, NaN) is numpy's"""
    dt = np.dtype(dt)
    if fill is None:
        fill = 0.0
    if not is_integer(dt):
        return float(fill)
    return np.pad(np.zeros(1, dtype=dt), (1, 0), "constant", constant_values=fill)[0]


def lane_of(dt) -> np.dtype:
    """the integer lanes an array whose result keeps its width computes on (diff / min / max / pad / gather / + - *):
    64-bit types on int64 lanes as they are, everything narrower on int32 lanes -- int32 / uint32 as they are, bool /
    8 / 16-bit widened by xg_convert.  Two's-complement arithmetic in the lane width followed by narrowing IS the narrow
    dtype's wrap-around."""
    dt = np.dtype(dt)
    if not is_integer(dt):
        raise TypeError(f"{dt} is not an integer dtype")
    return INT64 if dt.itemsize == 8 else INT32


def same_bits(dt, lane) -> bool:
    """an integer array that IS its lanes (int64 / uint64 on int64 lanes, int32 / uint32 on int32 lanes): no conversion"""
    dt, lane = np.dtype(dt), np.dtype(lane)
    return dt.kind in "iu" and dt.itemsize == lane.itemsize


class StencilPlan(NamedTuple):
    lanes: str                 # "int": *_i64 / *_i32 kernels; "float": convert the field first, *_f64 / *_f32 kernels
    compute: np.dtype          # float lanes: the float dtype; int lanes: int64 or int32 (lane_of)
    unsigned: bool             # min / max of an array that fills its lanes unsigned (uint64, uint32): XG_OP_MINU / MAXU
    result: np.dtype           # dtype of the operator's result BEFORE an output metric divides it
    via: Optional[np.dtype]    # int lanes: wrap the lane result to this dtype's width first (the narrow dtype's arithmetic)
    scale: float               # int lanes, interp: 0.5 applied in float64 after the conversion
    divide_as: Optional[np.dtype]  # int lanes with an output metric: the float dtype of `result / m_out`


def stencil_plan(op: str, x_dt, m_in_dt=None, m_out_dt=None) -> StencilPlan:
    """diff / interp / min / max of an array of dtype `x_dt` with optional metrics (reference order of operations:
    `da * m_in` first, xgcm/grid.py:804-808; the body; `/ m_out`, :830-832)."""
    x_dt = np.dtype(x_dt)
    metrics = [np.dtype(d) for d in (m_in_dt, m_out_dt) if d is not None]
    if not is_integer(x_dt) or m_in_dt is not None:
        f = float_of(x_dt, *metrics)  # `int * float_metric` promotes before anything else happens
        return StencilPlan("float", f, False, f, None, 1.0, None)
    if x_dt == BOOL and op == "diff":
        # a[..., 1:] - a[..., :-1] on booleans: numpy's own refusal
        raise TypeError("numpy boolean subtract, the `-` operator, is not supported, use the bitwise_xor, the `^` operator, "
                        "or the logical_xor function instead.")
    if op == "interp":  # (a[:-1] + a[1:]) in the array's dtype (wraps; bool: logical or), then / 2.0 -> float64
        result, via, scale = FLOAT64, x_dt, 0.5
    else:
        result, via, scale = x_dt, None, 1.0
    divide_as = None if m_out_dt is None else float_of(result, m_out_dt)
    lane = lane_of(x_dt)
    # narrower unsigned arrays are zero-extended into their lanes: the signed comparison is already right
    unsigned = x_dt.kind == "u" and same_bits(x_dt, lane) and op in ("min", "max")
    return StencilPlan("int", lane, unsigned, result, via, scale, divide_as)


def cumsum_dtype(x_dt) -> np.dtype:
    """numpy.cumsum / numpy.sum of integers: narrower types accumulate in the platform integer (int64 / uint64)"""
    x_dt = np.dtype(x_dt)
    if not is_integer(x_dt):
        return x_dt
    return UINT64 if x_dt.kind == "u" else INT64


def binary_plan(op: str, a_dt, b_dt):
    """`a OP b` of two arrays: ("int", result dtype) when numpy keeps the result integral, else ("float", float dtype)"""
    a_dt, b_dt = np.dtype(a_dt), np.dtype(b_dt)
    if is_integer(a_dt) and is_integer(b_dt) and op != "div":
        if a_dt == BOOL and b_dt == BOOL and op == "sub":
            raise TypeError("numpy boolean subtract, the `-` operator, is not supported, use the bitwise_xor, the `^` operator, "
                            "or the logical_xor function instead.")
        rt = np.result_type(a_dt, b_dt)
        if is_integer(rt):
            return "int", rt
        return "float", FLOAT64  # int64 with uint64: numpy goes to float64
    if is_integer(a_dt) and is_integer(b_dt):
        return "float", FLOAT64  # true division of integers
    return "float", float_of(a_dt, b_dt)
