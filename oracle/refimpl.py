"""CPU oracle for the xgcm Grid.diff/interp/min/max/cumsum/derivative/integrate hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``xgcm_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and only as the checker / the timed CPU baseline -- never as the thing shipped.

This is a numpy restatement of the exact numpy call sequence the reference bottoms out in
(the reference is pure Python over xarray -> numpy; xarray/dask are not installable in the
build container, so the xarray glue cannot be executed -- see DESIGN.md "Oracle").

Pinning status
--------------
* raw stencil bodies (`diff_forward`, `interp_forward`, `pairwise_forward_min/max`, the 8
  `cumsum_*` bodies): PINNED against outputs of the reference's own ``xgcm/gridops.py``
  functions executed in the build container (``oracle/make_golden.py`` ->
  ``tests/golden/gridops_vectors.npz``).
* `pad_basic`: the reference calls ``DataArray.pad(mode=wrap|constant|edge)`` which is
  ``numpy.pad``; numpy is available so the restatement *is* the third-party routine.
  Pinned by the known-answer tests transcribed from the reference's test-suite
  (``tests/golden/kats.json``).
* `grid_cumsum` trim/pad table, `derivative`, `integrate`: restated from reading
  ``xgcm/grid.py``; pinned by transcribed known-answer tests.  xarray's float-default
  ``skipna`` (NaN treated as 0 in ``cumsum``/``sum``) is xarray behaviour that no reference
  test exercises on this path: PARITY UNPINNED for NaN inputs to cumsum/integrate.

* the ``Grid`` LEVEL (dispatch, per-axis kwargs, the cumsum table, metric selection / interpolation, derivative / integrate /
  average / cumint, coordinate re-attachment, dim order, names, errors): PINNED MODULO A STAND-IN since round 5 --
  ``oracle/make_golden_grid.py`` runs the reference's own ``xgcm/grid.py`` stack over ``oracle/xr_min.py`` (numpy-backed
  stand-in for the xarray calls it makes) and ``tests/test_grid_reference.py`` replays its 426 calls through ``xgcm_amd.Grid``.
  ``oracle/make_golden_metadata.py`` does the same for ``Grid(ds)`` from COMODO / SGRID metadata (57 dataset descriptions,
  ``tests/test_metadata_reference.py``).
* the reference's OWN TEST SUITE and its OWN ``Grid``, live (round 5): ``oracle/run_reference_suite.py`` runs ``xgcm/test/*.py``,
  unmodified and read in place, against ``xgcm_amd`` (4032 passed on the oracle double, the same with deferred results, 3805 on
  the host build of the C ABI); ``oracle/fuzz_against_reference.py`` runs the reference's ``Grid`` and ``xgcm_amd.Grid`` side by
  side on seeded random grids / fields / calls and records the reference's answers as ``tests/golden/fuzz_reference.*``.  Both
  over the same numpy stand-in for xarray: PINNED MODULO THE STAND-IN.
* complex topologies (``oracle/topology.py``; fixtures ``fold_reference.json``, ``topology_reference.*``) and the vertical
  transform (``oracle/transform.py``; ``transform_kernels_reference.npz``): PINNED MODULO STAND-INS.  The halo logic and the
  two gufunc bodies that produced those fixtures are the reference's own code, loaded unmodified -- but over builder-written
  stand-ins for what the build container lacks: a numpy-backed ``DataArray`` for the dozen xarray operations
  ``_pad_face_connections`` / ``_pad_fold`` use (``oracle/make_golden_topology.py``) and a plain-Python ``guvectorize``
  (``oracle/make_golden_transform.py``).  What is pinned is the reference's LOGIC under the stand-ins' semantics of
  ``isel / concat / pad / transpose``; the transcribed known answers of the reference's own tests are the second,
  independent pin, and ``tests/test_real_xarray.py`` runs against real xarray (and an installed ``xgcm``) wherever they exist.

All functions take/return plain ``numpy.ndarray``; "axis" is an integer axis number of the
unlabelled array.  Metric arrays (``m_in``/``m_out``) must be numpy-broadcastable against
the input / output array (callers insert ``np.newaxis`` themselves).
"""

from __future__ import annotations

import numpy as np

# xgcm/padding.py:15-19
_XGCM_TO_NUMPY_PAD_MODE = {"periodic": "wrap", "fill": "constant", "extend": "edge"}

POSITIONS = ("center", "left", "right", "inner", "outer")

# xgcm/gridops.py:27-65 (diff), :80-117 (interp), :129-215 (min/max): (from, to) -> padding_width
STENCIL_PADDING_WIDTH = {
    ("center", "left"): (1, 0),
    ("left", "center"): (0, 1),
    ("center", "right"): (0, 1),
    ("right", "center"): (1, 0),
    ("center", "outer"): (1, 1),
    ("outer", "center"): (0, 0),
    ("center", "inner"): (0, 0),
    ("inner", "center"): (1, 1),
}


def position_length(n_center: int, pos: str) -> int:
    """Length of the dim at `pos` for an axis with `n_center` cells (docs/grids.md:77-79)."""
    return {"center": 0, "left": 0, "right": 0, "inner": -1, "outer": 1}[pos] + n_center


# --------------------------------------------------------------------------------------
# padding  (xgcm/padding.py:575-616 `_pad_basic`; one axis at a time, sequentially)
# --------------------------------------------------------------------------------------
def pad_basic(a: np.ndarray, axis: int, widths, padding, fill_value=0.0) -> np.ndarray:
    lo, hi = widths
    if lo == 0 and hi == 0:  # padding.py:592-593
        return a
    if padding is None:  # padding.py:601-608
        raise ValueError("No boundary condition was specified")
    mode = _XGCM_TO_NUMPY_PAD_MODE[padding]
    pw = [(0, 0)] * a.ndim
    pw[axis] = (int(lo), int(hi))
    if mode == "constant":  # padding.py:611-612
        return np.pad(a, pw, mode, constant_values=fill_value)
    return np.pad(a, pw, mode)


def pad_nd(a: np.ndarray, widths: dict, padding: dict, fill_value: dict) -> np.ndarray:
    """Sequential per-axis pad in dict order (padding.py:586-615). Keys are axis numbers."""
    out = a
    for axis, w in widths.items():
        out = pad_basic(out, axis, w, padding.get(axis), fill_value.get(axis, 0.0))
    return out


# --------------------------------------------------------------------------------------
# raw stencil bodies on the LAST axis of an already padded array (xgcm/gridops.py)
# --------------------------------------------------------------------------------------
def diff_forward(a):  # gridops.py:23-24
    return a[..., 1:] - a[..., :-1]


def interp_forward(a):  # gridops.py:76-77
    return (a[..., :-1] + a[..., 1:]) / 2.0


def pairwise_forward_min(a):  # gridops.py:123-126  (NaN-propagating np.min)
    return np.min(np.stack([a[..., :-1], a[..., 1:]], axis=-1), axis=-1)


def pairwise_forward_max(a):  # gridops.py:172-175
    return np.max(np.stack([a[..., :-1], a[..., 1:]], axis=-1), axis=-1)


_RAW = {
    "diff": diff_forward,
    "interp": interp_forward,
    "min": pairwise_forward_min,
    "max": pairwise_forward_max,
}


def stencil1d(
    op: str,
    a: np.ndarray,
    axis: int,
    pad_lo: int,
    pad_hi: int,
    padding,
    fill_value: float = 0.0,
    m_in: np.ndarray | None = None,
    m_out: np.ndarray | None = None,
) -> np.ndarray:
    """One axis of Grid.diff/interp/min/max exactly as the reference sequences it.

    grid.py:804-808  array = array * metric_in           (if metric_weighted)
    grid_ufunc.py:885-904  pad (copy) then apply:
        xr.apply_ufunc moves the core dim last (view), calls the raw body, and
        _restore_input_dim_order moves it back (grid_ufunc.py:56-103,954-990)
    grid.py:830-832  array = array / metric_out          (if metric_weighted / derivative)
    """
    if m_in is not None:
        a = a * m_in
    p = pad_basic(a, axis, (pad_lo, pad_hi), padding, fill_value)
    moved = np.moveaxis(p, axis, -1)
    r = _RAW[op](moved)
    out = np.moveaxis(r, -1, axis)
    if m_out is not None:
        out = out / m_out
    return np.ascontiguousarray(out)


# --------------------------------------------------------------------------------------
# the 8 registered cumsum grid-ufunc bodies (gridops.py:221-278)  [not used by Grid.cumsum]
# --------------------------------------------------------------------------------------
CUMSUM_UFUNC_TABLE = {
    # (from, to): (padding_width, pad_before_func, fill_value, trim_last)
    ("center", "left"): ((1, 0), False, 0, True),
    ("left", "center"): ((0, 0), True, None, False),
    ("center", "right"): ((0, 0), True, None, False),
    ("right", "center"): ((1, 0), False, 0, True),
    ("center", "outer"): ((1, 0), False, 0, False),
    ("outer", "center"): ((0, 0), True, None, True),
    ("center", "inner"): ((0, 0), True, None, True),
    ("inner", "center"): ((1, 0), False, 0, False),
}


def cumsum_ufunc_body(a: np.ndarray, trim_last: bool) -> np.ndarray:
    c = np.cumsum(a, axis=-1)
    return c[..., :-1] if trim_last else c


# --------------------------------------------------------------------------------------
# Grid.cumsum (xgcm/grid.py:1183-1418)
# --------------------------------------------------------------------------------------
def cumsum_trim_pad(from_pos: str, to_pos: str, reverse: bool):
    """Return (trim_lo, trim_hi, pad_lo, pad_hi) per grid.py:1326-1383."""
    natural = {("center", "right"), ("left", "center")}
    shifted = {("center", "left"), ("right", "center")}
    shrink = {("center", "inner"), ("outer", "center")}
    grow = {("center", "outer"), ("inner", "center")}
    pair = (from_pos, to_pos)
    if not reverse:
        if pair in natural:
            return 0, 0, 0, 0
        if pair in shifted:
            return 0, 1, 1, 0
        if pair in shrink:
            return 0, 1, 0, 0
        if pair in grow:
            return 0, 0, 1, 0
    else:
        if pair in shifted:
            return 0, 0, 0, 0
        if pair in natural:
            return 1, 0, 0, 1
        if pair in shrink:
            return 1, 0, 0, 0
        if pair in grow:
            return 0, 0, 0, 1
    raise ValueError(
        f"From `{from_pos}` to `{to_pos}` is not a valid position shift for cumsum"
    )


def cumsum1d(
    a: np.ndarray,
    axis: int,
    trim_lo: int,
    trim_hi: int,
    pad_lo: int,
    pad_hi: int,
    padding,
    fill_value: float = 0.0,
    reverse: bool = False,
    skipna: bool = True,
    m_in: np.ndarray | None = None,
    m_out: np.ndarray | None = None,
) -> np.ndarray:
    """One axis of Grid.cumsum.

    grid.py:1306-1308 `data * metric`; :1314-1318 flip / `DataArray.cumsum(dim)` / flip
    (xarray float default skipna -> numpy.nancumsum; PARITY UNPINNED for NaN);
    :1326-1383 trim;  :1385-1391 pad *the cumulative result*;  :1411-1414 `/ metric`.
    """
    if m_in is not None:
        a = a * m_in
    if reverse:
        a = np.flip(a, axis)
    c = np.nancumsum(a, axis=axis) if skipna else np.cumsum(a, axis=axis)
    if reverse:
        c = np.flip(c, axis)
    n = c.shape[axis]
    sl = [slice(None)] * c.ndim
    sl[axis] = slice(trim_lo, n - trim_hi)
    c = c[tuple(sl)]
    c = pad_basic(c, axis, (pad_lo, pad_hi), padding, fill_value)
    if m_out is not None:
        c = c / m_out
    return np.ascontiguousarray(c)


def grid_cumsum(a, axis, from_pos, to_pos, padding, fill_value=0.0, reverse=False, **kw):
    t = cumsum_trim_pad(from_pos, to_pos, reverse)
    return cumsum1d(a, axis, *t, padding, fill_value, reverse, **kw)


# --------------------------------------------------------------------------------------
# Grid.derivative (grid.py:1576-1578) and Grid.integrate (grid.py:1598-1605)
# --------------------------------------------------------------------------------------
def derivative(a, axis, pad_lo, pad_hi, padding, fill_value, dx_out):
    """`diff(da, axis) / get_metric(diff, (axis,))` -- metric at the OUTPUT position."""
    return stencil1d("diff", a, axis, pad_lo, pad_hi, padding, fill_value, m_out=dx_out)


def integrate(a: np.ndarray, axes, weight: np.ndarray | None, skipna: bool = True) -> np.ndarray:
    """`(da * weight).sum(dims)`; xarray float `sum` skips NaN (nansum == sum of NaN->0)."""
    w = a * weight if weight is not None else a
    if skipna:
        w = np.where(np.isnan(w), 0.0, w)
    if isinstance(axes, int):
        axes = (axes,)
    return np.sum(w, axis=tuple(axes))


def binary(op: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """xarray broadcasting arithmetic (`*`, `/`, `+`, `-`) on aligned numpy arrays."""
    return {"mul": np.multiply, "div": np.divide, "add": np.add, "sub": np.subtract}[op](a, b)


def vorticity(u, v, area, padding_x, padding_y, fill_x=0.0, fill_y=0.0):
    """Config 5 chain `(diff(v,'X') - diff(u,'Y')) / area` on (..., Y, X) arrays,
    both diffs center->left, i.e. padding_width (1,0) (docs/ufunc_examples.md vorticity)."""
    dvdx = stencil1d("diff", v, v.ndim - 1, 1, 0, padding_x, fill_x)
    dudy = stencil1d("diff", u, u.ndim - 2, 1, 0, padding_y, fill_y)
    return (dvdx - dudy) / area


def divergence(u, v, area, padding_x, padding_y, fill_x=0.0, fill_y=0.0):
    """`(diff(u,'X') + diff(v,'Y')) / area` on (..., Y, X) arrays, both diffs left->center, i.e.
    padding_width (0,1) (docs/ufunc_examples.md "Divergence", with the metric of the result)."""
    dudx = stencil1d("diff", u, u.ndim - 1, 0, 1, padding_x, fill_x)
    dvdy = stencil1d("diff", v, v.ndim - 2, 0, 1, padding_y, fill_y)
    return (dudx + dvdy) / area


def gradient(a, padding_x, padding_y, fill_x=0.0, fill_y=0.0, mx=None, my=None):
    """`(diff(a,'X') / mx, diff(a,'Y') / my)` on a (..., Y, X) array, both center->left
    (docs/ufunc_examples.md "Gradient"; with metrics: two `Grid.derivative` calls, grid.py:1465-1468)."""
    gx = stencil1d("diff", a, a.ndim - 1, 1, 0, padding_x, fill_x)
    gy = stencil1d("diff", a, a.ndim - 2, 1, 0, padding_y, fill_y)
    return (gx if mx is None else gx / mx), (gy if my is None else gy / my)


def flux(u, v, t, padding_x, padding_y, fill_x=0.0, fill_y=0.0):
    """`(u * interp(t,'X'), v * interp(t,'Y'))` on (..., Y, X) arrays, both interps center->left
    (docs/ufunc_examples.md "Advection": first-order advective flux of a tracer)."""
    tx = stencil1d("interp", t, t.ndim - 1, 1, 0, padding_x, fill_x)
    ty = stencil1d("interp", t, t.ndim - 2, 1, 0, padding_y, fill_y)
    return u * tx, v * ty


# --------------------------------------------------------------------------------------
# synthetic C-grid fields, bit-identical on host and device
# --------------------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (public-domain constant set)."""
    z = z.copy()
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def synthetic(n: int, seed: int, offset: int = 0, scale: float = 1.0, shift: float = -0.5) -> np.ndarray:
    """value(i) = u * scale + shift, u = (mix64(i + offset + seed*GOLDEN) >> 11) * 2**-53."""
    with np.errstate(over="ignore"):
        i = np.arange(offset, offset + n, dtype=np.uint64) + np.uint64(seed) * _GOLDEN
        u = (_mix64(i) >> np.uint64(11)).astype(np.float64) * (2.0**-53)
    return u * scale + shift


def synthetic_field(shape, seed: int) -> np.ndarray:
    return synthetic(int(np.prod(shape)), seed).reshape(shape)


def synthetic_metric(shape, seed: int) -> np.ndarray:
    """1000 * (1 + u): strictly positive, no zeros."""
    return synthetic(int(np.prod(shape)), seed, scale=1000.0, shift=1000.0).reshape(shape)
