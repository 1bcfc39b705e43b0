"""Test double for `xgcm_amd.device`, built on the CPU oracle.

TEST INFRASTRUCTURE ONLY.  The GPU-less `-m "not gpu"` suite uses it (via monkeypatch, see
tests/conftest.py::backend) to exercise the HOST logic of xgcm_amd -- Grid/Axis dispatch,
signature matching, kwarg precedence, coordinate re-attachment, error messages -- without
launching kernels.  The product never imports this module; on a GPU box the same tests run a
second time against the real HIP library.
"""

import numpy as np

from . import refimpl as R


def _nat(x):
    """the array as numpy holds its VALUES in native byte order (`astype`, numpy's own conversion -- never a view by dtype
    name): numpy computes a `>f8` / `>i4` array as it is and returns native results, so feeding the native twin to the
    oracle is the reference's arithmetic on the same values.  complex / object / datetime arrays are refused like the
    product refuses them."""
    if x is None:
        return None
    a = np.asarray(x)
    if a.dtype.kind not in "biuf" or a.dtype.itemsize > 8:
        kind = "complex arrays are" if a.dtype.kind == "c" else f"dtype {a.dtype} is"
        raise TypeError(f"array: {kind} not supported by the MI355X backend (served: bool, (u)int8-64, float16 / 32 / 64, "
                        "in either byte order)")
    return a if a.dtype.isnative else a.astype(a.dtype.newbyteorder("="))


def asdevice(x, dtype=None):
    """every served dtype keeps itself (floats -- float16 included -- and integers / bool: numpy computes them in their
    own dtype); mirrors device.asdevice"""
    a = _nat(x)
    return np.asarray(a, dtype=a.dtype if dtype is None else dtype, order="C")


def _common(*arrays):
    """the float lanes a mix of operands computes on: numpy's promotion, float32 only where it yields float32
    (mirrors device._common / xgcm_amd.dtypes.float_of); used by the FUSED operators only -- the reference-level
    operators below hand numpy the operands in their own dtypes and let IT promote, step by step"""
    present = [_nat(a).dtype for a in arrays if a is not None]
    rt = np.result_type(*present) if present else np.dtype(np.float64)
    return np.float32 if rt in (np.float32, np.float16) else np.float64


def _is_int(a):
    return a is not None and np.asarray(a).dtype.kind in "biu"


def _cast(dt, *arrays):
    return [None if a is None else asdevice(a, dt) for a in arrays]


def tohost(x):
    return np.asarray(x)


def is_device_array(x):
    return False


def stencil1d(op, x, axis, pad_lo, pad_hi, bc, fill=0.0, m_in=None, m_out=None):
    # numpy's own arithmetic in the operands' own dtypes is the oracle: integer wrap-around, fill cast, float16, and the
    # step-by-step promotion of `(x * m_in)` -> body -> `/ m_out` (a float32 difference is rounded before a float64 metric
    # divides it)
    x, m_in, m_out = _nat(x), _nat(m_in), _nat(m_out)
    return R.stencil1d(op, x, axis % x.ndim, pad_lo, pad_hi, bc, fill, m_in, m_out)


def cumsum1d(x, axis, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill=0.0, reverse=False, skipna=True, m_in=None, m_out=None):
    x, m_in, m_out = _nat(x), _nat(m_in), _nat(m_out)
    if _is_int(x) and m_in is None:
        skipna = False
    return R.cumsum1d(x, axis % x.ndim, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill, reverse, skipna, m_in, m_out)


def reduce1d(x, axis, w=None, skipna=True):
    x, w = _nat(x), _nat(w)
    if _is_int(x) and w is None and isinstance(skipna, (bool, int, np.bool_)):
        return np.sum(x, axis=axis % x.ndim)
    if x.dtype.kind in "biu" or (w is not None and np.result_type(x, w) != x.dtype):
        x = x.astype(np.result_type(x, w) if w is not None else np.float64)  # the dtype of `x * w`, for the count modes
    if skipna in ("pair_valid", "pair_all"):  # numerator and denominator sums stacked along a new leading dim
        valid = skipna == "pair_valid"
        return np.stack([reduce1d(x, axis, w, valid), reduce1d(x, axis, w, "valid" if valid else "all")])
    if skipna in ("mean_valid", "mean_all"):  # the one-pass weighted mean == the two sums, divided
        valid = skipna == "mean_valid"
        return reduce1d(x, axis, w, valid) / reduce1d(x, axis, w, "valid" if valid else "all")
    if skipna in ("valid", "all"):  # denominators of a weighted mean: weights of the valid / of all cells
        x = np.ones_like(x) if skipna == "all" else (~np.isnan(x)).astype(x.dtype)
        skipna = True
    return R.integrate(x, axis % x.ndim, w, skipna)


def pad_nd(x, widths, bc, fill):
    x = asdevice(x)
    return R.pad_nd(x, {k % x.ndim: v for k, v in widths.items()}, {k % x.ndim: v for k, v in bc.items()},
                    {k % x.ndim: (0.0 if v is None else v) for k, v in fill.items()})


def stencil1d_halo(op, x, halo, axis, pad_lo, pad_hi, m_out=None, m_in=None):
    x, halo, m_out, m_in = _nat(x), _nat(halo), _nat(m_out), _nat(m_in)
    if m_in is not None:  # `halo` holds the halo cells of the product already
        x = x * m_in
    halo = halo.astype(x.dtype)
    axis = axis % x.ndim
    lo = np.take(halo, range(0, pad_lo), axis=axis)
    hi = np.take(halo, range(pad_lo, pad_lo + pad_hi), axis=axis)
    return R.stencil1d(op, np.concatenate([lo, x, hi], axis=axis), axis, 0, 0, None, 0.0, None, m_out)


def transform_linear(phi, theta, target, axis, mask_edges=True, bypass_checks=False, logarithmic=False):
    from . import transform as TR

    phi, theta, target = _cast(_common(phi, theta, target), phi, theta, target)
    axis = axis % phi.ndim
    mv = lambda a: np.moveaxis(a, axis, -1)  # noqa: E731
    out = TR.interp_1d_linear(mv(phi), mv(theta), mv(target), mask_edges, bypass_checks, logarithmic)
    return np.ascontiguousarray(np.moveaxis(out, -1, axis))


def transform_conservative(phi, theta, bins, axis):
    from . import transform as TR

    phi, theta, bins = _cast(_common(phi, theta, bins), phi, theta, bins)
    axis = axis % phi.ndim
    out = TR.interp_1d_conservative(np.moveaxis(phi, axis, -1), np.moveaxis(theta, axis, -1), bins)
    return np.ascontiguousarray(np.moveaxis(out, -1, axis))


def upload_tokens(tokens):
    return np.ascontiguousarray(tokens, dtype=np.int64)


def gather(x, partner, tokens, mapped, lo, out_shape, fills, partner_perm=None):
    from . import topology as T

    if _is_int(x) and (partner is None or _is_int(partner)) and np.result_type(*[np.asarray(q).dtype for q in (x, partner) if q is not None]).kind in "biu":
        rt = np.result_type(*[np.asarray(q).dtype for q in (x, partner) if q is not None])
        x = np.asarray(x).astype(rt)
        partner = None if partner is None else np.asarray(partner).astype(rt)
        fills = [np.pad(np.zeros(1, dtype=rt), (1, 0), "constant", constant_values=f)[0] for f in fills]
    elif partner is not None:
        x, partner = _cast(_common(x, partner), x, partner)
    else:
        x = asdevice(x, _common(x))
    return T.gather_tokens(x, partner, tokens, mapped, lo, out_shape, fills, partner_perm)


def put_halo(out, halo, axis, pad_lo, pad_hi):
    axis = axis % out.ndim
    halo = np.asarray(halo).astype(out.dtype)
    idx = [slice(None)] * out.ndim
    for h in range(pad_lo + pad_hi):
        idx[axis] = h if h < pad_lo else out.shape[axis] - (pad_lo + pad_hi) + h
        out[tuple(idx)] = np.take(halo, h, axis=axis)
    return out


def binary(op, a, b):
    return R.binary(op, _nat(a), _nat(b))  # numpy's own promotion / wrap-around / true division / float16


def _with_halo(a, halo, axis, low, bc):
    """the array with its one-cell halo attached along `axis` (then no boundary mode is needed)"""
    if bc != "halo":
        return a, bc, (1, 0) if low else (0, 1)
    h = np.expand_dims(np.asarray(halo, dtype=a.dtype).reshape([n for i, n in enumerate(a.shape) if i != axis % a.ndim]), axis)
    return (np.concatenate([h, a], axis=axis) if low else np.concatenate([a, h], axis=axis)), None, (0, 0)


def vorticity(u, v, area, bc_x, bc_y, fill_x=0.0, fill_y=0.0, halo_x=None, halo_y=None):
    u, v, area = _cast(_common(u, v, area), u, v, area)
    if area is None:
        area = np.ones((1,) * u.ndim, dtype=u.dtype)
    if bc_x != "halo" and bc_y != "halo":
        return R.vorticity(u, v, area, bc_x, bc_y, fill_x, fill_y)
    vp, bx, px = _with_halo(v, halo_x, -1, True, bc_x)
    up, by, py = _with_halo(u, halo_y, -2, True, bc_y)
    dvdx = R.stencil1d("diff", vp, v.ndim - 1, px[0], px[1], bx, fill_x)
    dudy = R.stencil1d("diff", up, u.ndim - 2, py[0], py[1], by, fill_y)
    return (dvdx - dudy) / area


def divergence(u, v, area, bc_x, bc_y, fill_x=0.0, fill_y=0.0, halo_x=None, halo_y=None):
    u, v, area = _cast(_common(u, v, area), u, v, area)
    if area is None:
        area = np.ones((1,) * u.ndim, dtype=u.dtype)
    if bc_x != "halo" and bc_y != "halo":
        return R.divergence(u, v, area, bc_x, bc_y, fill_x, fill_y)
    up, bx, px = _with_halo(u, halo_x, -1, False, bc_x)
    vp, by, py = _with_halo(v, halo_y, -2, False, bc_y)
    dudx = R.stencil1d("diff", up, u.ndim - 1, px[0], px[1], bx, fill_x)
    dvdy = R.stencil1d("diff", vp, v.ndim - 2, py[0], py[1], by, fill_y)
    return (dudx + dvdy) / area


def gradient(a, bc_x, bc_y, fill_x=0.0, fill_y=0.0, mx=None, my=None, halo_x=None, halo_y=None):
    a, mx, my = _cast(_common(a, mx, my, halo_x, halo_y), a, mx, my)
    if bc_x != "halo" and bc_y != "halo":
        return R.gradient(a, bc_x, bc_y, fill_x, fill_y, mx, my)
    ax, bx, px = _with_halo(a, halo_x, -1, True, bc_x)
    ay, by, py = _with_halo(a, halo_y, -2, True, bc_y)
    gx = R.stencil1d("diff", ax, a.ndim - 1, px[0], px[1], bx, fill_x)
    gy = R.stencil1d("diff", ay, a.ndim - 2, py[0], py[1], by, fill_y)
    return (gx if mx is None else gx / mx), (gy if my is None else gy / my)


def flux(u, v, t, bc_x, bc_y, fill_x=0.0, fill_y=0.0, halo_x=None, halo_y=None):
    u, v, t = _cast(_common(u, v, t, halo_x, halo_y), u, v, t)
    if bc_x != "halo" and bc_y != "halo":
        return R.flux(u, v, t, bc_x, bc_y, fill_x, fill_y)
    tx, bx, px = _with_halo(t, halo_x, -1, True, bc_x)
    ty, by, py = _with_halo(t, halo_y, -2, True, bc_y)
    return (u * R.stencil1d("interp", tx, t.ndim - 1, px[0], px[1], bx, fill_x),
            v * R.stencil1d("interp", ty, t.ndim - 2, py[0], py[1], by, fill_y))


def stencil2d_supported(x, padx, pady):
    x = np.asarray(x)
    lane = 4 if x.dtype == np.float32 else 2
    return x.ndim >= 2 and x.shape[-1] % lane == 0 and sum(padx) == 1 and sum(pady) == 1 and x.shape[-1] > 0 and x.shape[-2] > 0


def stencil2d(op, x, order, padx, bc_x, fill_x, pady, bc_y, fill_y, metrics=None):
    x = asdevice(x)
    ax_x, ax_y = x.ndim - 1, x.ndim - 2
    m1 = m2 = m3 = None
    if metrics is not None:  # per axis: multiply by the metric at the current position, operate, divide at the new one
        m1, m2, m3 = (np.asarray(m).astype(x.dtype) for m in metrics)
    if order == 0:
        t = R.stencil1d(op, x, ax_x, padx[0], padx[1], bc_x, fill_x, m1, m2)
        return R.stencil1d(op, t, ax_y, pady[0], pady[1], bc_y, fill_y, m2, m3)
    t = R.stencil1d(op, x, ax_y, pady[0], pady[1], bc_y, fill_y, m1, m2)
    return R.stencil1d(op, t, ax_x, padx[0], padx[1], bc_x, fill_x, m2, m3)


def synthetic(shape, seed, offset=0, scale=1.0, shift=-0.5, out=None, dtype=np.float64):
    return R.synthetic(int(np.prod(shape)), seed, offset, scale, shift).reshape(tuple(shape)).astype(dtype)


_NAMES = ["asdevice", "tohost", "is_device_array", "stencil1d", "stencil1d_halo", "cumsum1d", "reduce1d", "pad_nd", "gather", "put_halo", "upload_tokens", "transform_linear", "transform_conservative", "binary",
          "vorticity", "divergence", "gradient", "flux", "stencil2d", "stencil2d_supported", "synthetic"]


def install(monkeypatch):
    import xgcm_amd.device as dev

    for n in _NAMES:
        monkeypatch.setattr(dev, n, globals()[n])
