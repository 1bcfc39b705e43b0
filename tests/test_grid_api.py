"""The public `Grid` surface against the reference's semantics.

Every test runs twice (fixture `backend`): on CPU with the oracle-backed device double (host
logic only, no kernels) and -- marked gpu -- through the real C ABI on an MI355X.  Tests mirror
the reference's own suites (file:line in each docstring) with seeded instead of unseeded data.
"""

import json
import os
import warnings

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid, apply_as_grid_ufunc, as_grid_ufunc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLDEN, "kats.json")) as f:
    KATS = json.load(f)


# ----------------------------------------------------------------------------------------------
# fixtures
# ----------------------------------------------------------------------------------------------
def grid_from_kat(kat, padding="__kat__"):
    spec = kat["grid"]
    coords = {}
    for ax, positions in spec["axes"].items():
        for pos, (dim, n) in positions.items():
            coords[dim] = (dim, np.arange(n, dtype=float))
    data_vars = {}
    metrics = None
    if "metrics" in spec:
        metrics = {}
        for ax, ms in spec["metrics"].items():
            metrics[(ax,)] = list(ms)
            for name, (dims, values) in ms.items():
                data_vars[name] = (tuple(dims), np.array(values, dtype=float))
    ds = Dataset(data_vars, coords)
    gcoords = {ax: {pos: dn[0] for pos, dn in positions.items()} for ax, positions in spec["axes"].items()}
    return Grid(ds, coords=gcoords, padding=spec["padding"] if padding == "__kat__" else padding, metrics=metrics,
                autoparse_metadata=False)


def second_order_diff(a):
    return 0.5 * (a[..., 2:] - a[..., :-2])


def cgrid(seed=100):
    """Seeded analogue of reference test/datasets.py:554-724 `datasets_grid_metric("C")`:
    dims (x, y, time, z) = (4, 5, 10, 6), center/right positions, metrics stored as coords."""
    nx, ny, nt, nz = 4, 5, 10, 6
    rnd = lambda shape, s: R.synthetic_field(shape, seed + s) + 0.5  # noqa: E731  (0,1) like np.random.rand
    coords = {
        "xt": ("xt", np.arange(nx) * 1.0), "xu": ("xu", np.arange(nx) + 0.5),
        "yt": ("yt", np.arange(ny) * 1.0), "yu": ("yu", np.arange(ny) + 0.5),
        "zt": ("zt", np.arange(nz) * 1.0), "zw": ("zw", np.arange(nz) + 0.5),
        "time": ("time", np.arange(nt) * 1.0),
    }
    one = np.ones((nx, ny))
    m = {
        "dx_ne": (("xu", "yu"), one * 0.3 - 0.1), "dx_n": (("xt", "yu"), one * 0.3 - 0.2),
        "dx_e": (("xu", "yt"), one * 0.3 - 0.25), "dx_t": (("xt", "yt"), one * 0.3 + 0.4),
        "dy_ne": (("xu", "yu"), one * 2 + 0.1), "dy_n": (("xt", "yu"), one * 2 + 0.2),
        "dy_e": (("xu", "yt"), one * 2 + 0.3), "dy_t": (("xt", "yt"), one * 2 + 0.4),
        "dz_t": (("xt", "yt", "time", "zt"), rnd((nx, ny, nt, nz), 1) * 20 + 1),
        "dz_w": (("xt", "yt", "time", "zw"), rnd((nx, ny, nt, nz), 2) * 20 + 1),
        "dz_w_e": (("xu", "yt", "time", "zw"), rnd((nx, ny, nt, nz), 3) * 20 + 1),
        "dz_w_n": (("xt", "yu", "time", "zw"), rnd((nx, ny, nt, nz), 4) * 20 + 1),
    }
    for k in ("ne", "n", "e", "t"):
        m[f"area_{k}"] = (m[f"dx_{k}"][0], m[f"dx_{k}"][1] * m[f"dy_{k}"][1] + 0.1)
    m["volume_t"] = (("xt", "yt", "time", "zt"), (m["dx_t"][1] * m["dy_t"][1])[:, :, None, None] * m["dz_t"][1] + 0.25)
    coords.update(m)
    data_vars = {
        "tracer": (("xt", "yt", "time", "zt"), rnd((nx, ny, nt, nz), 11)),
        "u": (("xu", "yt", "time", "zt"), rnd((nx, ny, nt, nz), 12)),
        "v": (("xt", "yu", "time", "zt"), rnd((nx, ny, nt, nz), 13)),
        "wt": (("xt", "yt", "time", "zw"), rnd((nx, ny, nt, nz), 14)),
    }
    ds = Dataset(data_vars, coords)
    gcoords = {"X": {"center": "xt", "right": "xu"}, "Y": {"center": "yt", "right": "yu"},
               "Z": {"center": "zt", "right": "zw"}}
    metrics = {
        ("X",): ["dx_t", "dx_n", "dx_e", "dx_ne"], ("Y",): ["dy_t", "dy_n", "dy_e", "dy_ne"],
        ("Z",): ["dz_t", "dz_w", "dz_w_n", "dz_w_e"], ("X", "Y"): ["area_t", "area_n", "area_e", "area_ne"],
        ("X", "Y", "Z"): ["volume_t"],
    }
    return ds, gcoords, metrics


def five_position_grid(n=9, padding="periodic", ax="X"):
    """reference test/test_grid_ufunc.py:216-268 `create_1d_test_grid`."""
    lengths = {"c": n, "g": n, "r": n, "i": n - 1, "o": n + 1}
    ds = Dataset(coords={f"{ax}_{k}": (f"{ax}_{k}", np.arange(m, dtype=float)) for k, m in lengths.items()})
    coords = {ax: {"center": f"{ax}_c", "left": f"{ax}_g", "right": f"{ax}_r", "inner": f"{ax}_i", "outer": f"{ax}_o"}}
    return Grid(ds, coords=coords, padding=padding, autoparse_metadata=False)


def _np(da):
    return da.values


# ----------------------------------------------------------------------------------------------
# known-answer tests transcribed from the reference
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kat", KATS, ids=lambda k: k["name"])
def test_reference_known_answers(backend, kat):
    grid = grid_from_kat(kat)
    da = DataArray(np.array(kat["in"], dtype=float), kat["in_dims"], name="fld")
    call = kat["call"]
    kw = dict(call["kwargs"])

    def run():
        if call["method"] == "apply_as_grid_ufunc":
            kw["padding_width"] = {k: tuple(v) for k, v in kw["padding_width"].items()}
            return apply_as_grid_ufunc(second_order_diff, da, axis=call["axis"], grid=grid, **kw)
        return getattr(grid, call["method"])(da, call["axis"], **kw)

    if "raises" in kat:
        with pytest.raises({"ValueError": ValueError}[kat["raises"]], match=kat["match"]):
            run()
        return
    res = run()
    assert list(res.dims) == kat["out_dims"]
    got = _np(res)
    if "out" in kat:
        want = np.array(kat["out"], dtype=float)
        assert got.shape == want.shape
        if kat.get("exact"):
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, atol=kat.get("atol", 0))
    else:
        np.testing.assert_allclose(got[..., -1], kat["out_last"], atol=kat["atol"])


def test_config1_plumbing_against_reference_fixture(backend):
    """BASELINE.json configs[0]: Grid.diff along X on a 128 x 64 periodic 2-D C-grid, float64 -- the
    whole stack (Grid -> dispatch -> fused ufunc -> device layer) against the output of the
    reference's own `diff_center_to_left` body (tests/golden/config1.npz)."""
    fx = np.load(os.path.join(GOLDEN, "config1.npz"))
    assert np.array_equal(fx["in"], R.synthetic_field((64, 128), 1))  # the committed input IS the seed-1 field
    ds = Dataset({"T": (("YC", "XC"), fx["in"])},
                 coords={"XC": ("XC", np.arange(128) + 0.5), "XG": ("XG", np.arange(128) * 1.0), "YC": ("YC", np.arange(64) * 1.0)})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, padding="periodic", autoparse_metadata=False)
    d = grid.diff(ds["T"], "X")
    assert d.dims == ("YC", "XG") and np.array_equal(d.values, fx["diff_X_center_to_left_periodic"])
    assert np.array_equal(grid.interp(ds["T"], "X").values, fx["interp_X_center_to_left_periodic"])
    assert np.array_equal(R.stencil1d("diff", fx["in"], 1, 1, 0, "periodic"), fx["diff_X_center_to_left_periodic"])


# ----------------------------------------------------------------------------------------------
# diff / interp / min / max through the Grid: positions x boundaries x axes of an N-D array
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("funcname", ["diff", "interp", "min", "max"])
def test_stencil_all_position_pairs(backend, funcname):
    grid = five_position_grid(9, padding=None)
    dim_of = grid.axes["X"].coords
    length = {"center": 9, "left": 9, "right": 9, "inner": 8, "outer": 10}
    for (f, t), (lo, hi) in R.STENCIL_PADDING_WIDTH.items():
        a = R.synthetic_field((3, length[f], 4), 7)
        da = DataArray(a, ("t", dim_of[f], "k"))
        for bc in ("periodic", "fill", "extend"):
            res = getattr(grid, funcname)(da, "X", to=t, padding=bc, fill_value=1.5)
            assert res.dims == ("t", dim_of[t], "k")  # input order kept, core dim renamed in place (GH#533)
            want = R.stencil1d(funcname, a, 1, lo, hi, bc, 1.5)
            assert np.array_equal(_np(res), want, equal_nan=True)
            assert res.shape[1] == length[t]
        if lo or hi:
            with pytest.raises(ValueError, match="No boundary condition was specified for axis 'X'"):
                getattr(grid, funcname)(da, "X", to=t)
        else:  # zero-width pads need no boundary condition (reference padding.py:592-593)
            getattr(grid, funcname)(da, "X", to=t)


def test_default_shift_and_invalid_shift(backend):
    grid = five_position_grid(9)
    a = DataArray(R.synthetic_field((9,), 1), ("X_c",))
    assert grid.diff(a, "X").dims == ("X_g",)  # center -> left first (FALLBACK_SHIFTS)
    g = DataArray(R.synthetic_field((9,), 1), ("X_g",))
    assert grid.interp(g, "X").dims == ("X_c",)
    with pytest.raises(NotImplementedError, match=r"Could not find any pre-defined diff grid ufuncs with signature \(X:left\)->\(X:right\)"):
        grid.diff(g, "X", to="right")
    with pytest.raises(KeyError, match="None of the DataArray's dims"):
        grid.diff(DataArray(np.zeros(3), ("q",)), "X")
    with pytest.raises(ValueError, match="keep_coords"):
        grid.diff(a, "X", keep_coords=True)
    with pytest.raises(TypeError, match="must be either a DataArray or Dictionary"):
        grid.diff(np.zeros(9), "X")


def test_multi_axis_and_per_axis_kwargs(backend):
    """reference grid.py:775-779: to / padding / fill_value are scalar-or-per-axis-dict."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, padding={"X": "periodic", "Y": "fill"}, autoparse_metadata=False)
    a = ds["tracer"]
    res = grid.interp(a, ["X", "Y"], fill_value={"Y": 3.0})
    step = R.stencil1d("interp", a.values, 0, 0, 1, "periodic")
    want = R.stencil1d("interp", step, 1, 0, 1, "fill", 3.0)
    assert res.dims == ("xu", "yu", "time", "zt")
    assert np.array_equal(_np(res), want)
    # per-call padding overrides the grid default (reference test_grid.py:843-892)
    res2 = grid.diff(a, "X", padding="extend")
    assert np.array_equal(_np(res2), R.stencil1d("diff", a.values, 0, 0, 1, "extend"))
    grid_e = Grid(ds, coords=coords, padding="extend", autoparse_metadata=False)
    assert grid_e.diff(a, "X").equals(res2)


def test_two_axes_fused_equals_sequential(backend):
    """Grid.interp/diff/min/max(da, [ax1, ax2]) on the last two dims runs as ONE launch and must give
    exactly what the reference's per-axis loop gives (xgcm/grid.py:800-832), coords included."""
    nz, ny, nx = 3, 10, 16
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0),
              "YC": ("YC", np.arange(ny) + 0.5), "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) * 1.0),
              "lonG": (("YG", "XG"), R.synthetic_field((ny, nx), 70)), "lonC": (("YC", "XC"), R.synthetic_field((ny, nx), 71))}
    ds = Dataset({"T": (("Z", "YC", "XC"), R.synthetic_field((nz, ny, nx), 72))}, coords)
    gcoords = {"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}}
    from xgcm_amd import device as dev_mod

    calls = []
    real = dev_mod.stencil2d
    for padding in ({"X": "periodic", "Y": "extend"}, "fill", {"X": "extend", "Y": "periodic"}):
        grid = Grid(ds, coords=gcoords, padding=padding, fill_value=None, autoparse_metadata=False)
        for fn in ("interp", "diff", "min", "max"):
            f = getattr(grid, fn)
            for axes in (["X", "Y"], ["Y", "X"]):
                dev_mod.stencil2d = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
                try:
                    fused = f(ds["T"], axes, fill_value={"X": 2.0, "Y": -3.0})
                finally:
                    dev_mod.stencil2d = real
                seq = f(f(ds["T"], axes[0], fill_value={"X": 2.0, "Y": -3.0}), axes[1], fill_value={"X": 2.0, "Y": -3.0})
                assert fused.dims == ("Z", "YG", "XG")
                assert fused.equals(seq), (padding, fn, axes)
                assert "lonG" in fused.coords and "lonC" not in fused.coords
    assert len(calls) == 3 * 4 * 2  # every two-axis call really took the fused path
    # not the last two dims / metric weighting / odd nx -> sequential path, same answer as before
    grid = Grid(ds, coords=gcoords, padding="periodic", autoparse_metadata=False)
    Tt = ds["T"].transpose("YC", "Z", "XC")
    Tt = DataArray(np.ascontiguousarray(Tt.values), Tt.dims)
    got = grid.interp(Tt, ["X", "Y"])
    want = R.stencil1d("interp", R.stencil1d("interp", Tt.values, 2, 1, 0, "periodic"), 0, 1, 0, "periodic")
    assert got.dims == ("YG", "Z", "XG") and np.array_equal(got.values, want)
    with pytest.raises(ValueError, match="No boundary condition was specified"):
        Grid(ds, coords=gcoords, autoparse_metadata=False).interp(ds["T"], ["X", "Y"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_two_axes_metric_weighted_fused_equals_sequential(backend, dtype):
    """`metric_weighted=("X", "Y")` on both axes (area-weighted interpolation to the corner points): ONE launch with the
    three area planes (input positions, between the axes, output positions) gives the bits of the reference's per-axis
    loop -- multiply before, divide after EACH axis (xgcm/grid.py:804-828)."""
    nz, ny, nx = 3, 10, 16
    m = lambda s: R.synthetic_metric((ny, nx), s).astype(dtype)  # noqa: E731
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0),
              "YC": ("YC", np.arange(ny) + 0.5), "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) * 1.0)}
    ds = Dataset({"T": (("Z", "YC", "XC"), R.synthetic_field((nz, ny, nx), 72).astype(dtype)),
                  "rA": (("YC", "XC"), m(1)), "rAw": (("YC", "XG"), m(2)), "rAs": (("YG", "XC"), m(3)), "rAz": (("YG", "XG"), m(4))}, coords)
    gcoords = {"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}}
    from xgcm_amd import device as dev_mod

    calls = []
    real = dev_mod.stencil2d
    fv = {"X": 2.0, "Y": -3.0}
    for padding in ({"X": "periodic", "Y": "extend"}, "fill", {"X": "extend", "Y": "periodic"}):
        grid = Grid(ds, coords=gcoords, padding=padding, metrics={("X", "Y"): ["rA", "rAw", "rAs", "rAz"]}, autoparse_metadata=False)
        for fn in ("interp", "diff"):
            f = getattr(grid, fn)
            for axes in (["X", "Y"], ["Y", "X"]):
                dev_mod.stencil2d = lambda *a, **k: (calls.append(k.get("metrics") is not None), real(*a, **k))[1]
                try:
                    fused = f(ds["T"], axes, fill_value=fv, metric_weighted=("X", "Y"))
                finally:
                    dev_mod.stencil2d = real
                seq = f(f(ds["T"], axes[0], fill_value=fv, metric_weighted=("X", "Y")), axes[1], fill_value=fv, metric_weighted=("X", "Y"))
                assert fused.dims == ("Z", "YG", "XG") and fused.values.dtype == dtype
                assert np.array_equal(fused.values, seq.values, equal_nan=True), (padding, fn, axes)
                # against the oracle written out: multiply, operate, divide -- twice
                a, b = ("rAw", "rAz") if axes[0] == "X" else ("rAs", "rAz")
                ax0, ax1 = (2, 1) if axes[0] == "X" else (1, 2)
                pad0 = padding if isinstance(padding, str) else padding[axes[0]]
                pad1 = padding if isinstance(padding, str) else padding[axes[1]]
                t = R.stencil1d(fn, ds["T"].values, ax0, 1, 0, pad0, dtype(fv[axes[0]]), ds["rA"].values[None], ds[a].values[None])
                want = R.stencil1d(fn, t, ax1, 1, 0, pad1, dtype(fv[axes[1]]), ds[a].values[None], ds[b].values[None])
                assert np.array_equal(fused.values, want, equal_nan=True), (padding, fn, axes)
    assert calls == [True] * (3 * 2 * 2)  # every call took the fused path with its three metric planes
    # a metric that is not a plain (Y, X) plane at one of the three positions -> the per-axis kernels, same bits
    ds2 = Dataset({"T": ds["T"], "dxC": (("XG",), R.synthetic_metric((nx,), 9).astype(dtype)), "dxT": (("XC",), R.synthetic_metric((nx,), 8).astype(dtype))}, coords)
    grid = Grid(ds2, coords=gcoords, padding="periodic", metrics={("X",): ["dxC", "dxT"]}, autoparse_metadata=False)
    got = grid.interp(ds2["T"], ["X", "Y"], metric_weighted="X")
    seq = grid.interp(grid.interp(ds2["T"], "X", metric_weighted="X"), "Y", metric_weighted="X")
    assert np.array_equal(got.values, seq.values)


def test_two_axes_integer_input_takes_the_sequential_path(backend):
    """signed-integer data through `diff(da, [X, Y])` must give the dtype and values of the two
    single-axis calls (the reference pads the INTEGER array: numpy.pad truncates a fractional fill value)."""
    ny, nx = 6, 8
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0),
              "YC": ("YC", np.arange(ny) + 0.5), "YG": ("YG", np.arange(ny) * 1.0)}
    a = np.arange(ny * nx, dtype=np.int64).reshape(ny, nx) * 3 % 17
    ds = Dataset({"n": (("YC", "XC"), a)}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                padding="fill", autoparse_metadata=False)
    fv = {"X": 1.0, "Y": 2.7}
    for fn in ("diff", "min", "max"):
        f = getattr(grid, fn)
        both = f(ds["n"], ["X", "Y"], fill_value=fv)
        seq = f(f(ds["n"], "X", fill_value=fv), "Y", fill_value=fv)
        assert both.values.dtype == seq.values.dtype == np.int64, fn
        assert np.array_equal(both.values, seq.values), fn
        px = np.pad(a, ((0, 0), (1, 0)), constant_values=1)
        want_x = {"diff": px[:, 1:] - px[:, :-1], "min": np.minimum(px[:, 1:], px[:, :-1]), "max": np.maximum(px[:, 1:], px[:, :-1])}[fn]
        py = np.pad(want_x, ((1, 0), (0, 0)), constant_values=2.7)  # numpy casts 2.7 -> 2 for an integer array
        want = {"diff": py[1:] - py[:-1], "min": np.minimum(py[1:], py[:-1]), "max": np.maximum(py[1:], py[:-1])}[fn]
        assert np.array_equal(both.values, want), fn


def test_integrate_with_a_weight_that_adds_dims(backend):
    """the broadcast fall-back of `integrate` (metric with a dim the data lacks) returns the same
    container type as the fused branch and skips NaN by default like xarray's float `sum`."""
    nz, ny = 4, 5
    coords = {"Z": ("Z", np.arange(nz) + 0.5), "Zl": ("Zl", np.arange(nz) * 1.0), "Y": ("Y", np.arange(ny) * 1.0)}
    dz = R.synthetic_metric((nz, ny), 5)
    ds = Dataset({"dz": (("Z", "Y"), dz)}, coords)
    grid = Grid(ds, coords={"Z": {"center": "Z", "left": "Zl"}}, padding="fill", metrics={("Z",): ["dz"]},
                autoparse_metadata=False)
    col = R.synthetic_field((nz,), 6)
    col[1] = np.nan
    out = grid.integrate(DataArray(col, ("Z",)), "Z")
    assert isinstance(out, DataArray) and out.dims == ("Y",)
    want = np.nansum(col[:, None] * dz, axis=0)
    assert np.allclose(out.values, want, rtol=1e-14, atol=0)


def test_multi_axis_integrate_and_average_with_separable_metrics(backend):
    """volume = area(Y, X) * thickness(Z): each factor rides in the reduction that removes its first dim, numerator
    and denominator of the mean travel side by side -- same numbers as the reference's `(da * metric).sum(dims)` /
    weighted mean (grid.py:1598-1605, :1662-1685) to rounding, NaN cells skipped"""
    nz, ny, nx = 4, 5, 6
    T = R.synthetic_field((nz, ny, nx), 61)
    T[1, 2, 3] = np.nan
    T[0, 0, :] = np.nan
    rA, drF, dxT = R.synthetic_metric((ny, nx), 62), R.synthetic_metric((nz,), 63), R.synthetic_metric((ny, nx), 64)
    ds = Dataset({"rA": (("YC", "XC"), rA), "drF": (("Z",), drF), "dxT": (("YC", "XC"), dxT)},
                 coords={"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0, "YC": np.arange(ny) + 0.5, "YG": np.arange(ny) * 1.0,
                         "Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding="fill", metrics={("X", "Y"): ["rA"], ("Z",): ["drF"], ("X",): ["dxT"]}, autoparse_metadata=False)
    da = DataArray(T, ("Z", "YC", "XC"), name="T")
    vol = rA[None] * drF[:, None, None]
    valid = ~np.isnan(T)
    T0 = np.where(valid, T, 0.0)
    cases = {
        ("X", "Y", "Z"): (vol, (0, 1, 2)), ("X", "Y"): (np.broadcast_to(rA, T.shape), (1, 2)),
        ("Z", "X"): (np.broadcast_to(drF[:, None, None] * dxT[None], T.shape), (0, 2)),
    }
    for axes, (w, red) in cases.items():
        got = grid.integrate(da, list(axes))
        assert got.dims == tuple(d for i, d in enumerate(da.dims) if i not in red)
        np.testing.assert_allclose(got.values, (T0 * w).sum(axis=red), rtol=1e-12)
        avg = grid.average(da, list(axes))
        with np.errstate(invalid="ignore"):
            want = (T0 * w).sum(axis=red) / (w * valid).sum(axis=red)
        np.testing.assert_allclose(avg.values, want, rtol=1e-12, equal_nan=True)
        # NaN-propagating flavours
        np.testing.assert_allclose(grid.integrate(da, list(axes), skipna=False).values, (T * w).sum(axis=red), rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(grid.average(da, list(axes), skipna=False).values, (T * w).sum(axis=red) / w.sum(axis=red),
                                   rtol=1e-12, equal_nan=True)
    # the public get_metric keeps xarray's dim order for a product; internally it is laid out like the weighted array
    assert grid.get_metric(da, ("X", "Y", "Z")).dims == ("YC", "XC", "Z")
    assert grid.get_metric(da, ("X", "Y", "Z"), _layout=da.dims).dims == ("Z", "YC", "XC")
    np.testing.assert_array_equal(grid.get_metric(da, ("X", "Y", "Z"), _layout=da.dims).values,
                                  np.transpose(grid.get_metric(da, ("X", "Y", "Z")).values, (2, 0, 1)))


def test_vector_component_dict_input(backend):
    ds, coords, _ = cgrid()
    grid = Grid(ds, coords=coords, padding="periodic", autoparse_metadata=False)
    res = grid.interp({"X": ds["u"]}, "X", to="center")
    assert np.array_equal(_np(res), R.stencil1d("interp", ds["u"].values, 0, 1, 0, "periodic"))
    with pytest.raises(ValueError, match="exactly one key/value pair"):
        grid.interp({"X": ds["u"], "Y": ds["v"]}, "X")
    with pytest.raises(ValueError, match="unknown axis"):
        grid.interp({"Q": ds["u"]}, "X")


# ----------------------------------------------------------------------------------------------
# coordinates (reference test_grid.py:571-756, test_grid_ufunc.py:743-858)
# ----------------------------------------------------------------------------------------------
def test_coords_reattached_from_grid_and_inputs(backend):
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, padding="periodic", autoparse_metadata=False)
    a = ds["tracer"]
    a = a.assign_coords({"time": ("time", np.arange(10) * 100.0), "label": ("yt", np.arange(5) * 2.0)})
    res = grid.diff(a, "X")
    assert np.array_equal(res.coords["xu"].values, ds["xu"].values)  # shifted dim: coord from the grid dataset
    assert "xt" not in res.coords  # coords on the old core dim are dropped
    assert np.array_equal(res.coords["time"].values, np.arange(10) * 100.0)  # user-modified non-core coord survives
    assert np.array_equal(res.coords["label"].values, np.arange(5) * 2.0)  # coord unknown to the grid survives
    assert "dy_e" in res.coords and res.coords["dy_e"].dims == ("xu", "yt")  # dataset coords on result dims
    assert "dx_t" not in res.coords


@pytest.mark.parametrize("funcname", ["diff", "interp", "min", "max", "integrate", "average", "cumsum", "cumint", "derivative"])
def test_keep_coords_all_operators(backend, funcname):
    """reference test_grid.py:571-611: result coords == its dim coords + every dataset coord that fits."""
    ds, coords, metrics = cgrid()
    ds = Dataset(ds.data_vars, {**ds.coords, "yt_bis": ("yt", ds["yt"].values), "xt_bis": ("xt", ds["xt"].values)})
    grid = Grid(ds, coords=coords, metrics=metrics, padding="periodic", autoparse_metadata=False)
    func = getattr(grid, funcname)
    for axis_name in grid.axes:
        result = func(ds["tracer"], axis_name)
        fits = [c for c in ds.coords if set(ds[c].dims).issubset(result.dims)]
        assert set(result.coords) == set(fits), (funcname, axis_name)


@pytest.mark.parametrize("funcname", ["interp", "diff", "cumsum"])
def test_preserve_input_noncore_coords(backend, funcname):
    """reference test_grid.py:648-756 (GH #496): coords the user changed on NON-core dims survive with their
    dtype; the shifted core dim's coord comes from the grid; coords on the old core dim disappear."""
    N = 8
    ds = Dataset({"v": (("time", "XC"), R.synthetic_field((N, N), 5))},
                 coords={"XC": ("XC", np.arange(N) + 0.5), "XG": ("XG", np.arange(N) * 1.0),
                         "time": ("time", np.arange(N) * 600.0), "t_label": ("time", np.arange(N).astype("int64")),
                         "xc_aux": ("XC", np.arange(N).astype("int64") * 10)})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, padding="periodic", autoparse_metadata=False)
    new_time = (np.arange(N) * 600 / 3600.0).astype(np.float32)
    new_t_label = (np.arange(N) + 100).astype(np.float32)
    v = ds["v"].assign_coords({"time": ("time", new_time), "t_label": ("time", new_t_label),
                               "xc_aux": ("XC", (np.arange(N) + 500).astype(np.float32))})
    out = grid.cumsum(v, "X", to="left") if funcname == "cumsum" else getattr(grid, funcname)(v, "X")
    assert out.coords["time"].dtype == np.float32 and np.array_equal(out.coords["time"].values, new_time)
    assert out.coords["t_label"].dtype == np.float32 and np.array_equal(out.coords["t_label"].values, new_t_label)
    assert np.array_equal(out.coords["XG"].values, ds["XG"].values)
    assert "XC" not in out.dims and "xc_aux" not in out.coords


def test_no_coords_dataset(backend):
    """reference test_grid.py:173-186: datasets without dimension coordinates work."""
    ds = Dataset({"c": ("xc", R.synthetic_field((8,), 1)), "g": ("xg", R.synthetic_field((8,), 2))})
    grid = Grid(ds, coords={"X": {"center": "xc", "left": "xg"}}, padding="periodic", autoparse_metadata=False)
    assert len(grid.diff(ds["c"], "X").coords) == 0
    assert len(grid.interp(ds["c"], "X").coords) == 0


# ----------------------------------------------------------------------------------------------
# metrics (reference test_metrics_ops.py, test_metrics.py)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("funcname", ["interp", "diff", "min", "max"])
@pytest.mark.parametrize("variable", ["tracer", "u", "v"])
@pytest.mark.parametrize("metric_weighted", ["X", ("Y",), ("X", "Y"), ["X", "Y"]])
def test_weighted_metric_bitwise(backend, funcname, variable, metric_weighted):
    """reference test_metrics_ops.py:35-64: metric_weighted == (x*m) -> op -> / m_new, `.equals`."""
    ds, coords, metrics = cgrid()
    for padding_init, padding in [("fill", "extend"), ({"X": "periodic", "Y": "fill"}, "fill")]:
        grid = Grid(ds, coords=coords, metrics=metrics, padding=padding_init, autoparse_metadata=False)
        func = getattr(grid, funcname)
        for axis in ("X", "Y"):
            metric = grid.get_metric(ds[variable], metric_weighted)
            expected_raw = func(ds[variable] * metric, axis, padding=padding)
            metric_new = grid.get_metric(expected_raw, metric_weighted)
            expected = expected_raw / metric_new
            new = func(ds[variable], axis, metric_weighted=metric_weighted, padding=padding)
            assert new.dims == expected.dims
            assert np.array_equal(_np(new), _np(expected), equal_nan=True)


@pytest.mark.parametrize("multi_axis", ["X", ["X"], ["X", "Y"], ("Y", "X")])
def test_weighted_metric_multi_axis(backend, multi_axis):
    """reference test_metrics_ops.py:66-95."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, autoparse_metadata=False)
    expected = ds["tracer"]
    for ax in multi_axis:
        expected = grid.interp(expected, ax, metric_weighted=("X", "Y"), padding="fill")
    new = grid.interp(ds["tracer"], multi_axis, metric_weighted=("X", "Y"), padding="fill")
    assert new.equals(expected)


def test_derivative_c_grid(backend):
    """reference test_metrics_ops.py:181-216: derivative == diff / dx at the OUTPUT position, bitwise."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, padding="periodic", autoparse_metadata=False)
    for var, axes, dxs in [("tracer", ["X", "Y", "Z"], ["dx_e", "dy_n", "dz_w"]),
                           ("u", ["X", "Y"], ["dx_t", "dy_ne"]), ("wt", ["X", "Z"], ["dx_e", "dz_t"])]:
        for ax, dx in zip(axes, dxs):
            if var == "wt" and ax == "X":
                continue  # no dx registered at (xu, yt, zw)-compatible dims other than dx_e: covered by tracer
            got = grid.derivative(ds[var], ax)
            want = grid.diff(ds[var], ax) / ds[dx].reset_coords(drop=True)
            assert got.dims == want.dims
            assert np.array_equal(_np(got), _np(want))


def test_derivative_with_metric_weighting_is_two_divisions(backend):
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, padding="periodic", autoparse_metadata=False)
    got = grid.derivative(ds["tracer"], "X", metric_weighted=("Y",))
    want = grid.diff(ds["tracer"], "X", metric_weighted=("Y",)) / ds["dx_e"].reset_coords(drop=True)
    assert np.array_equal(_np(got), _np(want))


def test_integrate_cumint_average_formulas(backend):
    """reference test_metrics_ops.py:256-398 (`_expected_result`)."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, padding="fill", autoparse_metadata=False)
    tr = ds["tracer"]
    for axis, mname, dims in zip(["X", "Y", "Z", ["X", "Y"], ["X", "Y", "Z"]],
                                 ["dx_t", "dy_t", "dz_t", "area_t", "volume_t"],
                                 [["xt"], ["yt"], ["zt"], ["xt", "yt"], ["xt", "yt", "zt"]]):
        metric = ds[mname].reset_coords(drop=True)
        new = grid.integrate(tr, axis)
        nums = tuple(tr.dims.index(d) for d in dims)
        mfull = np.broadcast_to(metric.transpose(*[d for d in tr.dims if d in metric.dims]).values[
            tuple(slice(None) if d in metric.dims else None for d in tr.dims)], tr.shape)
        want = (tr.values * mfull).sum(axis=nums)
        assert new.dims == tuple(d for d in tr.dims if d not in dims)
        np.testing.assert_allclose(_np(new), want, rtol=1e-12)
        if isinstance(axis, list):
            np.testing.assert_allclose(_np(grid.integrate(tr, tuple(axis))), want, rtol=1e-12)
        avg = grid.average(tr, axis)
        np.testing.assert_allclose(_np(avg), want / mfull.sum(axis=nums), rtol=1e-12)
    # a single STRIDED axis is bit-exact (sequential sum, like numpy); the contiguous axis (Z here) is a tree
    want = R.integrate(tr.values, 1, ds["dy_t"].values[:, :, None, None])
    assert np.array_equal(_np(grid.integrate(tr, "Y")), want)
    np.testing.assert_allclose(_np(grid.integrate(tr, "Z")), R.integrate(tr.values, 3, ds["dz_t"].values), rtol=1e-12)
    # cumint == cumsum(da * metric)
    for ax, mname in (("X", "dx_t"), ("Z", "dz_t")):
        got = grid.cumint(tr, ax, padding="fill")
        want = grid.cumsum(tr * ds[mname].reset_coords(drop=True), ax, padding="fill")
        assert got.equals(want)


def test_average_skips_missing(backend):
    """reference test_metrics_ops.py:98-121."""
    x = np.arange(10.0)
    ds = Dataset({"data": ("x", np.ones(10))}, coords={"x": ("x", x), "weights": ("x", np.ones(10) * 30)})
    grid = Grid(ds, coords={"X": {"center": "x"}}, metrics={"X": ["weights"]}, autoparse_metadata=False)
    expected = grid.average(ds["data"], "X")
    holed = np.ones(10)
    holed[6:8] = np.nan
    got = grid.average(DataArray(holed, ("x",)), "X")
    np.testing.assert_allclose(_np(got), _np(expected))
    np.testing.assert_allclose(_np(got), 1.0)


def test_get_metric_conditions(backend):
    """reference test_metrics.py:104-326 (conditions 1-4, warnings, errors)."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, padding="periodic", autoparse_metadata=False)
    # (1) exact axes at this position
    m = grid.get_metric(ds["u"], ("X",))
    assert np.array_equal(m.values, ds["dx_e"].values)
    m = grid.get_metric(ds["v"], ("Y", "X"))
    assert np.array_equal(m.values, ds["area_n"].values)
    # (3) product of sub-axis metrics when no ('X','Z') metric is registered
    m = grid.get_metric(ds["tracer"], ("X", "Z"))
    want = ds["dx_t"].values[:, :, None, None] * ds["dz_t"].values
    assert set(m.dims) == {"xt", "yt", "time", "zt"}
    assert np.array_equal(m.transpose("xt", "yt", "time", "zt").values, want)
    # (2) registered under exact axes but elsewhere -> interpolated with a warning
    grid2 = Grid(ds, coords=coords, metrics={("X",): ["dx_t"]}, padding="periodic", autoparse_metadata=False)
    with pytest.warns(UserWarning, match="being interpolated from metrics at dimensions"):
        m2 = grid2.get_metric(ds["u"], ("X",))
    assert m2.dims == ("xu", "yt")
    assert np.array_equal(m2.values, R.stencil1d("interp", ds["dx_t"].values, 0, 0, 1, "extend"))
    # errors
    with pytest.raises(KeyError, match="Unable to find any combinations of metrics"):
        Grid(ds, coords=coords, autoparse_metadata=False).get_metric(ds["u"], ("X",))
    with pytest.raises(KeyError, match="not compatible with grid axes"):
        grid.set_metrics(("Q",), "dx_t")
    with pytest.raises(KeyError, match="not found in dataset"):
        grid.set_metrics(("X",), "nope")
    with pytest.raises(ValueError, match="already assigned in metrics"):
        grid.set_metrics(("X",), "dx_t")
    grid.set_metrics(("X",), "dx_t", overwrite=True)


def test_get_metric_reference_scenarios(backend):
    """reference test_metrics.py:172-326, scenario by scenario (values via the operators themselves)."""
    ds, coords, metrics = cgrid()
    raw = lambda name: ds[name].reset_coords(drop=True)  # noqa: E731
    # test_get_metric_orig
    grid = Grid(ds, coords=coords, metrics=metrics, autoparse_metadata=False)
    for axes, var, want in [("X", "tracer", "dx_t"), (["X", "Y"], "tracer", "area_t"), (("X", "Y"), "tracer", "area_t"),
                            (["X", "Y", "Z"], "tracer", "volume_t"), (["X"], "u", "dx_e"), (["X", "Y"], "u", "area_e")]:
        assert np.array_equal(grid.get_metric(ds[var], axes).values, raw(want).values)
    # 02a: exact axes registered elsewhere -> interpolated (boundary 'extend'), default-shift semantics
    g = Grid(ds, coords=coords, padding="extend", autoparse_metadata=False)
    g.set_metrics(("X", "Y"), "area_e")
    with pytest.warns(UserWarning, match="being interpolated"):
        got = g.get_metric(ds["v"], ("X", "Y"))
    assert np.array_equal(got.values, g.interp(ds["area_e"], ("X", "Y")).values)
    # 02b: the exact-axes metric wins over sub-axis metrics even at matching positions
    g = Grid(ds, coords=coords, padding="periodic", autoparse_metadata=False)
    g.set_metrics(("X", "Y"), "area_e")
    g.set_metrics(("X"), "dx_n")
    g.set_metrics(("Y"), "dx_n")
    with pytest.warns(UserWarning, match="being interpolated"):
        got = g.get_metric(ds["v"], ("X", "Y"))
    assert np.array_equal(got.values, g.interp(ds["area_e"], ("X", "Y"), padding="extend").values)
    # 03a / 03b: products of sub-axis metrics
    g = Grid(ds, coords=coords, autoparse_metadata=False)
    g.set_metrics(("X"), "dx_n")
    g.set_metrics(("Y"), "dy_n")
    assert np.array_equal(g.get_metric(ds["v"], ("X", "Y")).values, ds["dx_n"].values * ds["dy_n"].values)
    g = Grid(ds, coords=coords, autoparse_metadata=False)
    g.set_metrics(("X", "Y"), "area_t")
    g.set_metrics(("Z"), "dz_t")
    m = g.get_metric(ds["tracer"], ("X", "Y", "Z")).transpose("xt", "yt", "time", "zt")
    assert np.array_equal(m.values, ds["area_t"].values[:, :, None, None] * ds["dz_t"].values)
    # 04a / 04b: wrong-position sub-axis metrics are interpolated first, one warning
    g = Grid(ds, coords=coords, padding="periodic", autoparse_metadata=False)
    g.set_metrics(("X"), "dx_t")
    g.set_metrics(("Y"), "dy_n")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = g.get_metric(ds["v"], ("X", "Y"))
    assert len([w for w in caught if "being interpolated" in str(w.message)]) == 1
    want = g.interp(ds["dx_t"], "Y", padding="extend").values * ds["dy_n"].values
    assert np.array_equal(got.values, want)
    g = Grid(ds, coords=coords, padding="periodic", autoparse_metadata=False)
    g.set_metrics(("X"), "dx_t")
    g.set_metrics(("Y"), "dy_t")
    with pytest.warns(UserWarning, match="being interpolated"):
        got = g.get_metric(ds["v"], ("X", "Y"))
    want = g.interp(ds["dx_t"], "Y", padding="extend").values * g.interp(ds["dy_t"], "Y", padding="extend").values
    assert np.array_equal(got.values, want)
    # GH #756: an exact-position combination found late must not emit interpolation warnings
    for var, zs, dz in (("tracer", ["dz_w", "dz_w_n", "dz_w_e", "dz_t"], "dz_t"), ("wt", ["dz_t", "dz_w"], "dz_w")):
        g = Grid(ds, coords=coords, metrics={("X", "Y"): ["area_t", "area_n", "area_e", "area_ne"], ("Z",): zs},
                 autoparse_metadata=False)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            m = g.get_metric(ds[var], ("X", "Y", "Z"))
        assert [w for w in caught if "being interpolated" in str(w.message)] == []
        dims = ds[dz].dims
        assert np.array_equal(m.transpose(*dims).values, ds["area_t"].values[:, :, None, None] * ds[dz].values)


# ----------------------------------------------------------------------------------------------
# cumsum (reference test_grid.py:196-370)
# ----------------------------------------------------------------------------------------------
def test_cumsum_positions_boundaries_reverse_nd(backend):
    grid = five_position_grid(9)
    dim_of = grid.axes["X"].coords
    length = {"center": 9, "left": 9, "right": 9, "inner": 8, "outer": 10}
    pairs = [("center", "left"), ("center", "right"), ("center", "outer"), ("center", "inner"), ("left", "center"),
             ("right", "center"), ("outer", "center"), ("inner", "center")]
    for f, t in pairs:
        a = R.synthetic_field((2, length[f], 3), 21)
        da = DataArray(a, ("t", dim_of[f], "k"))
        for bc in ("fill", "extend", "periodic"):
            for rev in (False, True):
                res = grid.cumsum(da, "X", to=t, padding=bc, fill_value=2.0, reverse=rev)
                want = R.grid_cumsum(a, 1, f, t, bc, 2.0, rev)
                assert res.dims == ("t", dim_of[t], "k")
                assert np.array_equal(_np(res), want)
                assert res.shape[1] == length[t]
    with pytest.raises(ValueError, match="is not a valid position shift for cumsum"):
        grid.cumsum(DataArray(np.zeros(9), ("X_g",)), "X", to="right", padding="fill")
    with pytest.raises(TypeError, match="unexpected keyword"):
        grid.cumsum(DataArray(np.zeros(9), ("X_c",)), "X", bogus=1)
    assert grid.cumsum(da, "X", padding="fill").equals(grid.cumsum(da, "X", padding="fill", reverse=False))


def test_cumsum_reverse_dict_and_metric(backend):
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, padding="fill", autoparse_metadata=False)
    tr = ds["tracer"]
    res = grid.cumsum(tr, ["X", "Z"], reverse={"Z": True})
    step = R.grid_cumsum(tr.values, 0, "center", "right", "fill", 0.0, False)
    want = R.grid_cumsum(step, 3, "center", "right", "fill", 0.0, True)
    # Z is the contiguous axis here: the GPU block scan re-associates the sum (1e-12 criterion)
    np.testing.assert_allclose(_np(res), want, rtol=1e-12, atol=1e-13)
    res = grid.cumsum(tr, ["X", "Y"], reverse={"Y": True})  # strided axes only: bit-exact
    assert np.array_equal(_np(res), R.grid_cumsum(step, 1, "center", "right", "fill", 0.0, True))
    with pytest.raises(ValueError, match="which are not being"):
        grid.cumsum(tr, "X", reverse={"Y": True})
    # metric_weighted: (x*m) -> cumsum -> / m_new (reference grid.py:1306-1308,1411-1414)
    got = grid.cumsum(tr, "X", metric_weighted=("X", "Y"))
    raw = grid.cumsum(tr * grid.get_metric(tr, ("X", "Y")), "X")
    want = raw / grid.get_metric(raw, ("X", "Y"))
    assert np.array_equal(_np(got), _np(want))


def test_cumsum_skips_nan_like_xarray(backend):
    """xarray's float cumsum is nancumsum (PARITY UNPINNED by the reference's tests; documented)."""
    grid = five_position_grid(9)
    a = R.synthetic_field((9,), 3)
    a[4] = np.nan
    res = grid.cumsum(DataArray(a, ("X_c",)), "X", to="right")
    assert np.array_equal(_np(res), np.nancumsum(a))


# ----------------------------------------------------------------------------------------------
# user grid ufuncs: the generic plugin path (reference test_grid_ufunc.py:300-899,1213-1273)
# ----------------------------------------------------------------------------------------------
def test_user_ufunc_generic_path(backend):
    grid = five_position_grid(9)
    a = R.synthetic_field((4, 9), 5)
    da = DataArray(a, ("t", "X_c"), coords={"t": ("t", np.arange(4.0))})

    def interp(x):
        return 0.5 * (x[..., :-1] + x[..., 1:])

    @as_grid_ufunc(signature="(X:center)->(X:left)", padding_width={"X": (1, 0)}, padding="fill", fill_value=10)
    def interp_center_to_left(x):
        return interp(x)

    res = interp_center_to_left(grid, da, axis=[["X"]])
    assert res.dims == ("t", "X_g")
    assert np.array_equal(_np(res), interp(np.pad(a, [(0, 0), (1, 0)], constant_values=10)))  # decorator fill (GH#652)
    res = interp_center_to_left(grid, da, axis=[["X"]], fill_value=1)  # call-time beats decorator
    assert np.array_equal(_np(res), interp(np.pad(a, [(0, 0), (1, 0)], constant_values=1)))
    res = interp_center_to_left(grid, da, axis=[["X"]], padding="periodic")
    assert np.array_equal(_np(res), interp(np.pad(a, [(0, 0), (1, 0)], mode="wrap")))
    assert np.array_equal(res.coords["t"].values, np.arange(4.0))
    # dim order of a non-last core dim is restored (GH#533)
    dat = DataArray(np.ascontiguousarray(a.T), ("X_c", "t"))
    res = interp_center_to_left(grid, dat, axis=[["X"]])
    assert res.dims == ("X_g", "t")
    assert np.array_equal(_np(res), interp(np.pad(a, [(0, 0), (1, 0)], constant_values=10)).T)
    # wrong trimming is reported like the reference does
    bad = as_grid_ufunc(signature="(X:center)->(X:left)", padding_width={"X": (1, 1)}, padding="fill")(interp)
    with pytest.raises(ValueError, match="does your grid ufunc correctly trim"):
        bad(grid, da.assign_coords({"X_c": ("X_c", np.arange(9.0))}), axis=[["X"]])
    with pytest.raises(ValueError, match="does not appear in argument"):
        interp_center_to_left(grid, DataArray(a, ("t", "X_g")), axis=[["X"]])
    with pytest.raises(ValueError, match="Must provide an axis"):
        apply_as_grid_ufunc(interp, da, grid=grid, signature="(X:center)->(X:left)")
    with pytest.raises(ValueError, match="Must provide a grid"):
        apply_as_grid_ufunc(interp, da, axis=[["X"]], signature="(X:center)->(X:left)")


def test_user_ufunc_two_axes_two_outputs(backend):
    """reference test_grid_ufunc.py:601-660 (gradient to inner points, multiple returns)."""
    lengths = {"c": 9, "i": 8}
    coords = {}
    for ax, n in (("lon", 9), ("lat", 11)):
        coords[f"{ax}_c"] = (f"{ax}_c", np.arange(n) * 1.0)
        coords[f"{ax}_i"] = (f"{ax}_i", np.arange(n - 1) + 0.5)
    grid = Grid(Dataset(coords=coords), coords={"lon": {"center": "lon_c", "inner": "lon_i"},
                                                "lat": {"center": "lat_c", "inner": "lat_i"}},
                padding="periodic", autoparse_metadata=False)
    a = R.synthetic_field((9, 11), 8)
    da = DataArray(a, ("lon_c", "lat_c"))

    @as_grid_ufunc(signature="(X:center,Y:center)->(X:inner,Y:center),(X:center,Y:inner)")
    def grad_to_inner(x):
        return x[..., 1:, :] - x[..., :-1, :], x[..., 1:] - x[..., :-1]

    u, v = grad_to_inner(grid, da, axis=[("lon", "lat")])
    assert u.dims == ("lon_i", "lat_c") and v.dims == ("lon_c", "lat_i")
    assert np.array_equal(_np(u), a[1:, :] - a[:-1, :])
    assert np.array_equal(_np(v), a[:, 1:] - a[:, :-1])


def test_builtin_ufunc_objects_are_callable_like_the_reference(backend):
    """gridops.<name>.ufunc is the raw body on padded unlabelled arrays; calling the object is fused."""
    from xgcm_amd import gridops

    p = np.pad(np.arange(10.0) ** 2, (1, 0), "wrap")
    assert np.array_equal(gridops.diff_center_to_left.ufunc(p), [-81, 1, 3, 5, 7, 9, 11, 13, 15, 17])
    p = np.pad(np.linspace(1, 10, 10), (1, 1), "edge")
    assert np.array_equal(gridops.interp_center_to_outer.ufunc(p), np.concatenate(([1.0], np.linspace(1.5, 9.5, 9), [10.0])))
    grid = five_position_grid(9)
    sq = np.arange(1, 10.0) ** 2
    res = gridops.cumsum_center_to_left(grid, DataArray(sq, ("X_c",)), axis=[("X",)], padding="fill")
    assert np.array_equal(_np(res), np.concatenate([[0.0], np.cumsum(sq)[:-1]]))  # decorator fill_value=0
    res = gridops.cumsum_center_to_outer(grid, DataArray(sq, ("X_c",)), axis=[("X",)], padding="extend")
    assert np.array_equal(_np(res), np.concatenate([[sq[0]], np.cumsum(sq)]))


def test_fused_vorticity_equals_operator_chain(backend):
    nz, ny, nx = 3, 6, 8
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0),
              "YC": ("YC", np.arange(ny) + 0.5), "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) * 1.0)}
    ds = Dataset({"U": (("Z", "YC", "XG"), R.synthetic_field((nz, ny, nx), 51)),
                  "V": (("Z", "YG", "XC"), R.synthetic_field((nz, ny, nx), 52)),
                  "rAz": (("YG", "XG"), R.synthetic_metric((ny, nx), 53))}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                metrics={("X", "Y"): ["rAz"]}, padding="fill", autoparse_metadata=False)
    zeta = grid.vorticity(ds["U"], ds["V"])
    chain = (grid.diff(ds["V"], "X") - grid.diff(ds["U"], "Y")) / ds["rAz"].reset_coords(drop=True)
    assert zeta.dims == ("Z", "YG", "XG")
    assert np.array_equal(_np(zeta), _np(chain))
    assert np.array_equal(_np(zeta), R.vorticity(ds["U"].values, ds["V"].values, ds["rAz"].values[None], "fill", "fill"))


def test_fused_divergence_equals_operator_chain(backend):
    """docs/ufunc_examples.md "Divergence": u (YC, XG), v (YG, XC) -> cell centre, one launch."""
    nz, ny, nx = 3, 6, 8
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0),
              "YC": ("YC", np.arange(ny) + 0.5), "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) * 1.0)}
    ds = Dataset({"U": (("Z", "YC", "XG"), R.synthetic_field((nz, ny, nx), 61)),
                  "V": (("Z", "YG", "XC"), R.synthetic_field((nz, ny, nx), 62)),
                  "rA": (("YC", "XC"), R.synthetic_metric((ny, nx), 63))}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                metrics={("X", "Y"): ["rA"]}, padding={"X": "periodic", "Y": "fill"}, autoparse_metadata=False)
    div = grid.divergence(ds["U"], ds["V"])
    chain = (grid.diff(ds["U"], "X") + grid.diff(ds["V"], "Y")) / ds["rA"].reset_coords(drop=True)
    assert div.dims == ("Z", "YC", "XC")
    assert np.array_equal(_np(div), _np(chain))
    assert np.array_equal(_np(div), R.divergence(ds["U"].values, ds["V"].values, ds["rA"].values[None], "periodic", "fill"))
    # the user-ufunc form of the docs (pad (0,1) on both axes, then trimmed differences) gives the same bits
    def divergence(u, v):
        return (u[..., :-1, 1:] - u[..., :-1, :-1]) + (v[..., 1:, :-1] - v[..., :-1, :-1])

    ref = grid.apply_as_grid_ufunc(divergence, ds["U"], ds["V"], axis=[("Y", "X"), ("Y", "X")],
                                   signature="(Y:center,X:left),(Y:left,X:center)->(Y:center,X:center)",
                                   padding_width={"X": (0, 1), "Y": (0, 1)})
    assert np.array_equal(_np(grid.divergence(ds["U"], ds["V"], metric_weighted=False)), _np(ref))
    with pytest.raises(NotImplementedError, match="u at"):
        grid.divergence(ds["V"], ds["U"])


def test_fused_gradient_and_flux_equal_operator_chains(backend):
    """docs/ufunc_examples.md "Gradient" and "Advection": one centre field in, two staggered fields out."""
    nz, ny, nx = 3, 6, 8
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0),
              "YC": ("YC", np.arange(ny) + 0.5), "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) * 1.0)}
    ds = Dataset({"T": (("Z", "YC", "XC"), R.synthetic_field((nz, ny, nx), 60)),
                  "U": (("Z", "YC", "XG"), R.synthetic_field((nz, ny, nx), 61)),
                  "V": (("Z", "YG", "XC"), R.synthetic_field((nz, ny, nx), 62)),
                  "dxC": (("YC", "XG"), R.synthetic_metric((ny, nx), 63)),
                  "dyC": (("YG", "XC"), R.synthetic_metric((ny, nx), 64))}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                metrics={("X",): ["dxC"], ("Y",): ["dyC"]}, padding={"X": "periodic", "Y": "extend"},
                autoparse_metadata=False)
    gx, gy = grid.gradient(ds["T"])
    assert gx.dims == ("Z", "YC", "XG") and gy.dims == ("Z", "YG", "XC")
    assert np.array_equal(_np(gx), _np(grid.diff(ds["T"], "X")))
    assert np.array_equal(_np(gy), _np(grid.diff(ds["T"], "Y")))
    gx, gy = grid.gradient(ds["T"], metric_weighted=True, padding={"Y": "fill"}, fill_value={"Y": 2.5})
    assert np.array_equal(_np(gx), _np(grid.derivative(ds["T"], "X")))
    assert np.array_equal(_np(gy), _np(grid.derivative(ds["T"], "Y", padding="fill", fill_value=2.5)))
    ex, ey = R.gradient(ds["T"].values, "periodic", "fill", 0.0, 2.5, ds["dxC"].values[None], ds["dyC"].values[None])
    assert np.array_equal(_np(gx), ex) and np.array_equal(_np(gy), ey)

    # the user-ufunc form of the docs (pad (1,0) on both axes, trimmed differences) gives the same bits
    def gradient(a):
        return a[..., 1:, 1:] - a[..., 1:, :-1], a[..., 1:, 1:] - a[..., :-1, 1:]

    rx, ry = grid.apply_as_grid_ufunc(gradient, ds["T"], axis=[("Y", "X")],
                                      signature="(Y:center,X:center)->(Y:center,X:left),(Y:left,X:center)",
                                      padding_width={"X": (1, 0), "Y": (1, 0)})
    gx, gy = grid.gradient(ds["T"])
    assert np.array_equal(_np(gx), _np(rx)) and np.array_equal(_np(gy), _np(ry))

    fx, fy = grid.flux(ds["U"], ds["V"], ds["T"])
    assert fx.dims == ("Z", "YC", "XG") and fy.dims == ("Z", "YG", "XC")
    assert np.array_equal(_np(fx), _np(ds["U"] * grid.interp(ds["T"], "X")))
    assert np.array_equal(_np(fy), _np(ds["V"] * grid.interp(ds["T"], "Y")))
    ex, ey = R.flux(ds["U"].values, ds["V"].values, ds["T"].values, "periodic", "extend")
    assert np.array_equal(_np(fx), ex) and np.array_equal(_np(fy), ey)
    # the advect() step of the docs: T - dt * div(flux)
    adv = ds["T"] - 3.0 * grid.divergence(fx, fy, metric_weighted=False)
    ref = ds["T"].values - 3.0 * R.divergence(ex, ey, np.ones((1, 1, 1)), "periodic", "extend")
    assert np.array_equal(_np(adv), ref)
    # (X, Y)-ordered dims: the two operator calls run instead, same labelled result
    tt = ds["T"].transpose("Z", "XC", "YC")
    gx2, gy2 = grid.gradient(tt)
    assert np.array_equal(_np(gx2.transpose("Z", "YC", "XG")), _np(gx))
    with pytest.raises(NotImplementedError, match="center"):
        grid.gradient(ds["U"])
    with pytest.raises(NotImplementedError, match="tracer"):
        grid.flux(ds["U"], ds["V"], ds["U"])


def test_grid_constructor_errors(backend):
    ds, coords, metrics = cgrid()
    with pytest.raises(ValueError, match="`periodic` argument has been removed"):
        Grid(ds, coords=coords, periodic=False, autoparse_metadata=False)
    with pytest.raises(ValueError, match="renamed to 'padding'"):
        Grid(ds, coords=coords, boundary="fill", autoparse_metadata=False)
    with pytest.raises(TypeError, match="unexpected keyword"):
        Grid(ds, coords=coords, bogus=1, autoparse_metadata=False)
    with pytest.raises(ValueError, match="Could not determine Axis names"):
        Grid(ds, autoparse_metadata=False)
    with pytest.raises(TypeError, match="must be of type xarray.Dataset"):
        Grid(np.zeros(3), coords=coords)
    with pytest.warns(DeprecationWarning, match="default fill_value will be changed"):
        g = Grid(ds, coords=coords, fill_value={"X": 1.0}, autoparse_metadata=False)
    assert g.axes["X"].fill_value == 1.0 and g.axes["Y"].fill_value == 0.0
    assert repr(g).split("\n")[0] == "<xgcm.Grid>"


def test_integer_inputs_keep_their_dtype_like_numpy(backend):
    """diff / min / max / cumsum of signed integers return integers (numpy's result), interp returns
    floats; a non-integral fill value is truncated as numpy.pad does."""
    grid = five_position_grid(9, padding="fill")
    a = (np.arange(18).reshape(2, 9) ** 2 % 17 - 5).astype(np.int64)
    da = DataArray(a, ("t", "X_c"))
    for fn in ("diff", "min", "max"):
        got = getattr(grid, fn)(da, "X", to="left", fill_value=2.7)
        want = R.stencil1d(fn, a, 1, 1, 0, "fill", 2.7)   # numpy on the int array: pad casts 2.7 -> 2
        assert _np(got).dtype == np.int64 and want.dtype == np.int64
        np.testing.assert_array_equal(_np(got), want)
    got = grid.interp(da, "X", to="left", fill_value=2.7)
    assert _np(got).dtype == np.float64
    np.testing.assert_array_equal(_np(got), R.stencil1d("interp", a, 1, 1, 0, "fill", 2.7))
    got = grid.cumsum(da, "X", to="left", padding="fill")
    assert _np(got).dtype == np.int64
    np.testing.assert_array_equal(_np(got), np.concatenate([np.zeros((2, 1), np.int64), np.cumsum(a, axis=1)[:, :-1]], axis=1))
    a32 = a.astype(np.int32)
    got = grid.diff(DataArray(a32, ("t", "X_c")), "X", to="right", padding="periodic")
    assert _np(got).dtype == np.int32
    np.testing.assert_array_equal(_np(got), np.roll(a32, -1, axis=1) - a32)


def test_sharded_core_axis_single_rank_equals_grid_methods(backend):
    """xgcm_amd.sharding.stencil_along_sharded_axis with one rank (no exchange): the halo-mode kernel fed
    with locally made boundary planes gives the bits of the ordinary in-kernel boundary modes."""
    from xgcm_amd.sharding import stencil_along_sharded_axis

    nz, ny, nx = 6, 5, 8
    ds = Dataset({"T": (("Z", "YC", "XC"), R.synthetic_field((nz, ny, nx), 21))},
                 {"Z": ("Z", np.arange(nz) * 1.0), "Zl": ("Zl", np.arange(nz) - 0.5), "XC": ("XC", np.arange(nx) + 0.5),
                  "XG": ("XG", np.arange(nx) * 1.0)})
    for bc in ("periodic", "fill", "extend"):
        grid = Grid(ds, coords={"Z": {"center": "Z", "left": "Zl"}, "X": {"center": "XC", "left": "XG"}}, padding=bc,
                    fill_value=2.5, autoparse_metadata=False)
        for fn in ("diff", "interp", "min"):
            for ax in ("Z", "X"):
                got = stencil_along_sharded_axis(grid, fn, ds["T"], ax)
                want = getattr(grid, fn)(ds["T"], ax)
                assert got.dims == want.dims and np.array_equal(_np(got), _np(want)), (bc, fn, ax)
    with pytest.raises(NotImplementedError, match="length-preserving"):
        g5 = five_position_grid(9, "fill")
        stencil_along_sharded_axis(g5, "diff", DataArray(np.arange(9.0), ("X_c",)), "X", to="outer")
    from xgcm_amd.sharding import cumsum_along_sharded_axis

    grid = Grid(ds, coords={"Z": {"center": "Z", "left": "Zl"}, "X": {"center": "XC", "left": "XG"}}, padding="extend",
                autoparse_metadata=False)
    for ax in ("Z", "X"):  # one rank: exactly Grid.cumsum
        got = cumsum_along_sharded_axis(grid, ds["T"], ax)
        want = grid.cumsum(ds["T"], ax)
        assert got.dims == want.dims and np.array_equal(_np(got), _np(want))
