"""Vertical coordinate transform (reference xgcm/transform.py, Grid.transform).

* `tests/golden/transform_cases.json` is the `cases` table of the reference's own test-suite
  (xgcm/test/test_transform.py:40-686, inputs + expected outputs, written by oracle/make_golden.py);
  it pins the oracle (oracle/transform.py) and, through `Grid.transform`, the product;
* the reference's low / mid / high level tests (:849-1424) are mirrored with file:line references;
* seeded sweeps (NaN-laden, decreasing, duplicated theta; broadcasting; float32) compare the HIP
  kernels with the oracle: bit-exact for linear and conservative, 1e-12 for method="log" (libm).

Product tests run twice via `backend`: CPU with the oracle-backed device double, GPU through the C ABI.
"""

import json
import os
import warnings

import numpy as np
import pytest

from oracle import refimpl as R
from oracle import transform as TR
from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd import transform as X

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLDEN, "transform_cases.json")) as f:
    CASES = json.load(f)

ALL = list(CASES)
NOT_CONS_MULTIDIM = [c for c in ALL if not ("conservative" in c and "multidim_target" in c)]
LINEAR_1D = [c for c in ALL if ("linear" in c or "log" in c) and "multidim_target" not in c]
CONS_1D = [c for c in ALL if "conservative" in c and "multidim_target" not in c]
MULTIDIM = ["conservative_depth_dens_nonmono_edge", "linear_depth_dens", "linear_depth_depth", "conservative_depth_temp"]


def _arr(v):
    return np.array(v, dtype=float)


def construct(case):
    """numpy/duck-array form of reference test_transform.py:688-769 `construct_test_source_data`."""
    case = dict(case)

    def make_da(prefix):
        name, data = case[prefix + "_data"]
        cname, cvals = case[prefix + "_coord"]
        if prefix + "_dims" in case:
            dims = tuple(case[prefix + "_dims"])
            coords = {cname: (dims, _arr(cvals))}
        else:
            dims = (cname,)
            coords = {cname: _arr(cvals)}
        return DataArray(_arr(data), dims=dims, coords=coords, name=name)

    def make_ds(prefix):
        da = make_da(prefix)
        ds = Dataset({da.name: da})
        if prefix + "_additional_data" in case and prefix + "_additional_data_coord" in case:
            an, av = case[prefix + "_additional_data"]
            cn, cv = case[prefix + "_additional_data_coord"]
            ds[an] = DataArray(_arr(av), dims=[cn], coords={cn: _arr(cv)}, name=an)
        if prefix + "_bounds_coord" in case:
            bn, bv = case[prefix + "_bounds_coord"]
            ds[bn] = (bn, _arr(bv))
        if prefix + "_data_mask_index" in case:
            vals = ds[da.name].values.copy()
            for ii in case[prefix + "_data_mask_index"]:
                vals[tuple(ii) if isinstance(ii, list) else ii] = np.nan
            ds[da.name] = DataArray(vals, dims=da.dims, coords=dict(da.coords), name=da.name)
        return ds

    source, expected, target = make_ds("source"), make_ds("expected"), make_da("target")
    kw = dict(case["transform_kwargs"])
    if kw.get("target_data") is not None:
        kw["target_data"] = source[kw["target_data"]].copy()
    return source, dict(case["grid_kwargs"]), target, kw, expected, case.get("error")


def _expected(expected, kw):
    return expected["data" + kw.get("suffix", "")]


def _assert_like_reference(got, want):
    """xr.testing.assert_allclose of the reference (rtol 1e-5, NaN == NaN, same dims)."""
    assert tuple(got.dims) == tuple(want.dims)
    np.testing.assert_allclose(got.values, want.values, rtol=1e-5, atol=1e-8, equal_nan=True)
    for name, c in want.coords.items():
        np.testing.assert_allclose(np.asarray(got.coords[name].values, dtype=float), np.asarray(c.values, dtype=float), rtol=1e-5)


# ----------------------------------------------------------------------------------------------
# the oracle against the reference's table (CPU)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", LINEAR_1D)
def test_oracle_linear_matches_reference_cases(name):
    source, gkw, target, kw, expected, _ = construct(CASES[name])
    theta = kw["target_data"].values if kw.get("target_data") is not None else source[source.data.dims[0]].values
    out = TR.interp_1d_linear(source.data.values, theta, target.values, mask_edges=kw.get("mask_edges", True),
                              logarithmic=kw["method"] == "log")
    np.testing.assert_allclose(out, _expected(expected, kw).values, rtol=1e-5, equal_nan=True)


@pytest.mark.parametrize("name", [c for c in CONS_1D if not CASES[c].get("error")])
def test_oracle_conservative_matches_reference_cases(name):
    source, gkw, target, kw, expected, _ = construct(CASES[name])
    bounds_dim = gkw["coords"]["Z"]["outer"]
    theta = kw["target_data"].values if kw.get("target_data") is not None else source[bounds_dim].values
    out = TR.interp_1d_conservative(source.data.values, theta, target.values)
    np.testing.assert_allclose(out, _expected(expected, kw).values, rtol=1e-5, equal_nan=True)
    np.testing.assert_allclose(np.nansum(out), np.nansum(source.data.values))  # test_transform.py:1043


# ----------------------------------------------------------------------------------------------
# low level (test_transform.py:849-921)
# ----------------------------------------------------------------------------------------------
def test_interp_1d_linear(backend):
    """:849-864: uniformly stratified scalar, analytic answer."""
    nz, nx = 100, 1000
    zv = np.linspace(0, 1, nz + 1)
    z = 0.5 * (zv[:-1] + zv[1:])
    x = 2 * np.pi * np.linspace(0, 1, nx)
    theta = z + 0.1 * np.cos(3 * x)[:, None]
    phi = np.sin(theta) + 0.1 * np.cos(5 * x)[:, None]
    levels = np.arange(0.2, 0.9, 0.025)
    got = X.interp_1d_linear(phi, theta, levels, mask_edges=False)
    np.testing.assert_allclose(got, np.sin(levels) + 0.1 * np.cos(5 * x)[:, None], rtol=1e-4)
    np.testing.assert_array_equal(got, TR.interp_1d_linear(phi, theta, levels, mask_edges=False))


def test_interp_1d_conservative(backend):
    """:867-898: the column integral is conserved; NaNs in the data are ignored; bad targets raise."""
    nz = 30
    dz = 10 + np.linspace(0, 90, nz - 1)
    z = np.concatenate([[0], np.cumsum(dz)])
    H = z.max()
    theta = z / H + 0.2 * np.cos(np.pi * z / H)
    bins = np.linspace(theta.min() - 0.1, theta.max() + 0.1, 100)
    dz2, th2 = np.tile(dz, (5, 1)), np.tile(theta, (5, 1))
    out = X.interp_1d_conservative(dz2, th2, bins)
    np.testing.assert_allclose(np.nansum(out, axis=-1), np.nansum(dz2, axis=-1))
    np.testing.assert_array_equal(out, TR.interp_1d_conservative(dz2, th2, bins))
    phi = np.array([1, 2, np.nan])
    np.testing.assert_allclose(X.interp_1d_conservative(phi, np.array([30.0, 40, 50, 60]), np.array([30.0, 50])), np.nansum(phi))
    with pytest.raises(ValueError):
        X.interp_1d_conservative(dz2, th2, np.array([0.0, -2, 4]))


# ----------------------------------------------------------------------------------------------
# mid level (test_transform.py:923-1046)
# ----------------------------------------------------------------------------------------------
def test_mid_level_rejects_plain_arrays(backend):
    """:923-947."""
    source, _, target, _, _, _ = construct(CASES["linear_depth_depth"])
    with pytest.raises(ValueError):
        X.linear_interpolation(source.data, source["z"], target.values, "z", "z", "z")
    source, _, target, _, _, _ = construct(CASES["conservative_depth_depth"])
    with pytest.raises(ValueError):
        X.conservative_interpolation(source.data, source["z"], target.values, "z", "z", "z")


@pytest.mark.parametrize("name", LINEAR_1D)
def test_mid_level_linear(backend, name):
    """:950-992."""
    source, _, target, kw, expected, error = construct(CASES[name])
    kw.setdefault("suffix", "")
    method = kw.pop("method")
    sdim, tdim = source.data.dims[0], target.dims[0]
    theta = kw.pop("target_data", None)
    if theta is None:
        theta = source[sdim]
    out = X.linear_interpolation(source.data, theta, target, sdim, sdim, tdim, logarithmic=(method == "log"), **kw)
    _assert_like_reference(out, expected["data" + kw["suffix"]])
    assert out.name == "data" + kw["suffix"]


@pytest.mark.parametrize("name", CONS_1D)
def test_mid_level_conservative(backend, name):
    """:995-1046 (the two cases flagged `error` there need the high-level interp and are skipped here)."""
    source, gkw, target, kw, expected, error = construct(CASES[name])
    if error:
        pytest.skip("mid level cannot handle this case in the reference either (xfail there)")
    kw.setdefault("suffix", "")
    kw.pop("method")
    sdim, bdim, tdim = gkw["coords"]["Z"]["center"], gkw["coords"]["Z"]["outer"], target.dims[0]
    theta = kw.pop("target_data", None)
    if theta is None:
        theta = source[bdim]
    out = X.conservative_interpolation(source.data, theta, target, sdim, bdim, tdim, **kw)
    _assert_like_reference(out, expected["data" + kw["suffix"]])
    np.testing.assert_allclose(np.nansum(out.values), np.nansum(source.data.values))
    assert out.name == "data" + kw["suffix"]


# ----------------------------------------------------------------------------------------------
# high level (test_transform.py:1052-1424)
# ----------------------------------------------------------------------------------------------
def _grid(source, gkw, **extra):
    return Grid(source, **{**gkw, **extra})


@pytest.mark.parametrize("name", NOT_CONS_MULTIDIM)
def test_grid_transform(backend, name):
    """:1052-1068: every case of the table through Grid.transform."""
    source, gkw, target, kw, expected, _ = construct(CASES[name])
    grid = _grid(source, gkw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        out = grid.transform(source.data, "Z", target, **kw)
    _assert_like_reference(out, _expected(expected, kw))


def test_conservative_multidim_target_and_explicit_target_dim(backend):
    """:1071-1105."""
    source, gkw, target, kw, _, _ = construct(CASES["conservative_depth_depth_multidim_target"])
    with pytest.raises(NotImplementedError, match="multi-dimensional targets"):
        _grid(source, gkw).transform(source.data, "Z", target, **kw)
    source, gkw, target, kw, expected, _ = construct(CASES["conservative_depth_depth_rename"])
    (tdim,) = target.dims
    assert len(tdim) > 1
    out = _grid(source, gkw).transform(source.data, "Z", target, target_dim=tdim, **kw)
    _assert_like_reference(out, _expected(expected, kw))


def test_conservative_warns_without_cell_bounds(backend):
    """:1108-1126."""
    source, gkw, target, kw, _, _ = construct(CASES["conservative_depth_temp"])
    with pytest.warns(UserWarning, match="The `target data` input is not located on the cell bounds"):
        _grid(source, gkw).transform(source.data, "Z", target, **kw)


@pytest.mark.parametrize("name", MULTIDIM)
def test_names_errors_and_auto_naming(backend, name):
    """:1129-1214: unnamed input, periodic axis, dimension naming for ndarray targets."""
    source, gkw, target, kw, expected, _ = construct(CASES[name])
    grid = _grid(source, gkw)
    unnamed = DataArray(source.data.values, dims=source.data.dims, coords=dict(source.data.coords))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        assert grid.transform(unnamed, "Z", target, **kw).name is None
        with pytest.raises(ValueError, match="non-periodic"):
            _grid(source, gkw, padding="periodic").transform(source.data, "Z", target, **kw)
        td = kw.get("target_data")
        if td is None:
            want = grid.axes["Z"].coords["center" if kw["method"] == "linear" else "outer"]
        else:
            want = td.name
        out = grid.transform(source.data, "Z", target.values, **kw)
        assert want in out.coords and want in out.dims


def test_noname_target_data_warns(backend):
    """:1146-1173."""
    source, gkw, target, kw, _, _ = construct(CASES["linear_depth_dens"])
    td = kw.pop("target_data")
    td = DataArray(td.values, dims=td.dims, coords=dict(td.coords))  # no name
    with pytest.warns(UserWarning, match="TRANSFORMED_DIMENSION"):
        out = _grid(source, gkw).transform(source.data, "Z", target.values, target_data=td, **kw)
    assert "TRANSFORMED_DIMENSION" in out.dims


@pytest.mark.parametrize("bypass", [True, False])
def test_bypass_checks(backend, bypass):
    """:1217-1245."""
    source, gkw, target, kw, expected, _ = construct(CASES["linear_depth_dens"])
    td = kw.pop("target_data")
    out = _grid(source, gkw).transform(source.data, "Z", target, target_data=td, bypass_checks=bypass, **kw)
    _assert_like_reference(out, expected["data"])


@pytest.mark.parametrize("name", MULTIDIM)
def test_grid_transform_multidim(backend, name):
    """:1285-1316: the column broadcast against another dim gives the 1-D answer everywhere."""
    source, gkw, target, kw, expected, _ = construct(CASES[name])
    na = 8
    col = source.data
    big = DataArray(np.repeat(col.values[None], na, axis=0), dims=("a",) + col.dims, coords=dict(col.coords), name="data")
    ds = Dataset({k: v for k, v in source.variables.items() if k != "data"})
    ds["data"] = big
    td = kw.pop("target_data", None)
    if td is not None:
        td = DataArray(np.repeat(td.values[None], na, axis=0), dims=("a",) + td.dims, coords=dict(td.coords), name=td.name)
    grid = _grid(ds, gkw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        out = grid.transform(big, "Z", target, target_data=td, **kw)
    want = _expected(expected, kw)
    assert out.dims == ("a",) + tuple(want.dims)
    np.testing.assert_allclose(out.values, np.broadcast_to(want.values, out.shape), rtol=1e-5, equal_nan=True)


@pytest.mark.parametrize("name", ["linear_depth_depth_nomask_multidim_target", "linear_depth_depth_multidim_target"])
def test_grid_transform_spatially_varying_target(backend, name):
    """:1319-1339: a 2-D target (eta_rho, s_rho) for a 1-D column."""
    source, gkw, target, kw, expected, _ = construct(CASES[name])
    td = kw.pop("target_data", None)
    out = _grid(source, gkw).transform(source.data, "Z", target, target_data=td, **kw)
    _assert_like_reference(out, expected["data"])


def test_other_dims_error_and_input_check(backend):
    """:1342-1368 and :1391-1424."""
    source, gkw, target, kw, _, _ = construct(CASES["linear_depth_dens"])
    grid = _grid(source, gkw)
    td = kw.pop("target_data")
    a3 = DataArray(np.repeat(source.data.values[None], 3, axis=0), dims=("a",) + source.data.dims, name="data")
    td_other = DataArray(np.repeat(td.values[None], 3, axis=0), dims=("a_other",) + td.dims, name=td.name)
    with pytest.raises(ValueError, match="additional dimensions"):
        grid.transform(a3, "Z", target, target_data=td_other, **kw)
    with pytest.raises(ValueError, match=r"`da` needs to be a"):
        grid.transform(source, "Z", target, target_data=td, **kw)
    with pytest.raises(ValueError, match="needs to be a"):
        grid.transform(source.data, "Z", Dataset({"dummy": target}), target_data=td, **kw)
    with pytest.raises(ValueError, match="needs to be a"):
        grid.transform(source.data, "Z", target, target_data=Dataset({"dummy": td}), **kw)
    no_outer = Grid(source, coords={"Z": {"center": "depth"}}, autoparse_metadata=False)
    with pytest.raises(RuntimeError, match="`outer` coordinates"):
        no_outer.transform(source.data, "Z", target, method="conservative", target_data=td)


# ----------------------------------------------------------------------------------------------
# seeded sweeps: kernels vs oracle on hard columns
# ----------------------------------------------------------------------------------------------
def _columns(shape, seed, kind):
    """theta fields (..., n): increasing / decreasing / with duplicates / NaN head, tail, holes."""
    rng = np.random.default_rng(seed)
    base = np.cumsum(rng.random(shape) + 0.05, axis=-1)
    if kind == "decreasing":
        base = base[..., ::-1].copy()
    if kind == "duplicates":
        base[..., 2] = base[..., 1]
        base[..., -1] = base[..., -2]
    if kind == "nan_tail":
        base[..., -2:] = np.nan
    if kind == "nan_head":
        base[..., :2] = np.nan
    if kind == "nan_holes":
        base[..., 3] = np.nan
        base[1:, ..., 0] = np.nan
    if kind == "nonmonotonic":
        base = base + 3 * np.sin(base)
    return base


@pytest.mark.parametrize("kind", ["increasing", "decreasing", "duplicates", "nan_tail", "nan_head", "nan_holes", "nonmonotonic"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_linear_kernel_equals_numpy_interp(backend, kind, dtype):
    shape = (3, 5, 17)
    theta = _columns(shape, 3, kind).astype(dtype)
    phi = (R.synthetic_field(shape, 5) * 10).astype(dtype)
    phi[0, 0, 4] = np.nan
    levels = np.concatenate([[-1.0, np.nan], np.linspace(0.0, float(np.nanmax(theta)) + 1, 23), [theta[1, 2, 5]]]).astype(dtype)
    # sorted, NaN-free targets (the kernel's streaming path on well-formed columns): values left and right
    # of every column, repeated levels, levels that coincide with theta entries (first, inner, last)
    clean = np.sort(np.concatenate([levels[~np.isnan(levels)], [theta[0, 0, 0], theta[2, 4, -1], theta[1, 2, 5]],
                                    np.nan_to_num(theta[2, 3, 6:9], nan=1.0)])).astype(dtype)
    for lv in (levels, clean):
        for mask in (False, True):
            for bypass in (False, True):
                got = X.interp_1d_linear(phi, theta, lv, mask_edges=mask, bypass_checks=bypass)
                want = TR.interp_1d_linear(phi, theta, lv, mask_edges=mask, bypass_checks=bypass)
                assert got.dtype == dtype
                np.testing.assert_array_equal(got, want)
    # every column interpolated onto its own (sorted) theta values: exact hits all the way
    own = np.sort(np.nan_to_num(theta, nan=0.5), axis=-1)
    np.testing.assert_array_equal(X.interp_1d_linear(phi, theta, own, mask_edges=True),
                                  TR.interp_1d_linear(phi, theta, own, mask_edges=True))
    # a target that varies from column to column, theta a single shared profile
    lv2 = (levels[None, None, 2:] + R.synthetic_field((3, 5, 1), 9)).astype(dtype)
    got = X.interp_1d_linear(phi, theta[0, 0], lv2, mask_edges=True)
    np.testing.assert_array_equal(got, TR.interp_1d_linear(phi, theta[0, 0], lv2, mask_edges=True))
    pos = np.abs(theta) + dtype(0.5)
    lv = np.abs(levels[2:]) + dtype(0.25)
    got = X.interp_1d_linear(phi, pos, lv, mask_edges=True, logarithmic=True)
    np.testing.assert_allclose(got, TR.interp_1d_linear(phi, pos, lv, mask_edges=True, logarithmic=True),
                               rtol=1e-12 if dtype == np.float64 else 5e-3, atol=0 if dtype == np.float64 else 1e-4,
                               equal_nan=True)  # float32: 1-ulp log differences, amplified by steep slopes


KERNEL_REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "transform_kernels_reference.npz"))


def _kernel_ref_cases(dn, kind):
    """(what, args, reference output) for every recorded call of the reference's own kernel bodies
    (oracle/make_golden_transform.py: xgcm/transform.py loaded unmodified, its numba gufuncs run column by column)."""
    g = lambda k: KERNEL_REF[f"{dn}/{kind}/{k}"]  # noqa: E731
    phi = KERNEL_REF[f"{dn}/phi"]
    for ln in ("levels", "clean"):
        for mask in (False, True):
            for bypass in (False, True):
                key = f"{dn}/{kind}/{ln}/linear/m{int(mask)}b{int(bypass)}"
                if key in KERNEL_REF.files:
                    yield "linear", (phi, g("theta"), g(ln)), dict(mask_edges=mask, bypass_checks=bypass), KERNEL_REF[key]
    if f"{dn}/{kind}/log/linear" in KERNEL_REF.files:
        yield "log", (phi, g("log_theta"), g("log_levels")), dict(mask_edges=True, logarithmic=True), g("log/linear")
    for bn in ("inc", "dec", "fine"):
        ref = g(f"conservative/{bn}")
        if bn == "dec":  # the reference flips axis 0 of the N-D result (transform.py:190-192); here the bin axis
            ref = ref[::-1][..., ::-1]
        yield "conservative", (phi, g("theta_outer"), g(f"bins_{bn}")), {}, ref


@pytest.mark.parametrize("kind", ["increasing", "decreasing", "duplicates", "nan_tail", "nan_head", "nan_holes", "nonmonotonic"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_transform_kernels_equal_reference_kernel_outputs(backend, kind, dtype):
    """Oracle AND product against the outputs of the reference's own `_interp_1d_linear` / `_interp_1d_conservative`
    bodies on the hard columns (NaN head / tail / holes, duplicates, non-monotonic), both dtypes."""
    dn = np.dtype(dtype).name
    n = 0
    for what, args, kw, ref in _kernel_ref_cases(dn, kind):
        n += 1
        assert ref.dtype == dtype
        if what == "conservative":
            np.testing.assert_array_equal(TR.interp_1d_conservative(*args), ref)
            np.testing.assert_array_equal(X.interp_1d_conservative(*args), ref)
        elif what == "log" and dtype == np.float32:  # log in double, rounded once (DESIGN.md deviations): tolerance
            for f in (TR.interp_1d_linear, X.interp_1d_linear):
                np.testing.assert_allclose(f(*args, **kw), ref, rtol=5e-3, atol=1e-4, equal_nan=True)
        elif what == "log":
            np.testing.assert_array_equal(TR.interp_1d_linear(*args, **kw), ref)
            np.testing.assert_allclose(X.interp_1d_linear(*args, **kw), ref, rtol=1e-12, atol=0, equal_nan=True)  # device log: 1 ulp
        else:
            np.testing.assert_array_equal(TR.interp_1d_linear(*args, **kw), ref)
            np.testing.assert_array_equal(X.interp_1d_linear(*args, **kw), ref)
    assert n >= 4


@pytest.mark.parametrize("kind", ["increasing", "decreasing", "duplicates", "nan_tail", "nan_head", "nan_holes", "nonmonotonic"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_conservative_kernel_equals_oracle(backend, kind, dtype):
    shape = (2, 4, 13)
    theta = _columns(shape[:-1] + (shape[-1] + 1,), 7, kind).astype(dtype)
    phi = (R.synthetic_field(shape, 8) * 100).astype(dtype)
    phi[1, 1, 3] = np.nan
    bins = np.linspace(-0.5, float(np.nanmax(theta)) + 0.5, 21).astype(dtype)   # 20 bins: 2.5 register tiles
    got = X.interp_1d_conservative(phi, theta, bins)
    assert got.dtype == dtype and got.shape == shape[:-1] + (20,)
    np.testing.assert_array_equal(got, TR.interp_1d_conservative(phi, theta, bins))
    np.testing.assert_array_equal(X.interp_1d_conservative(phi, theta, bins[::-1].copy()),
                                  TR.interp_1d_conservative(phi, theta, bins[::-1].copy()))


@pytest.mark.parametrize("kind", ["increasing", "decreasing", "nonmonotonic", "nan_holes", "thick"])
@pytest.mark.parametrize("nbins", [17, 40, 64, 65])
def test_conservative_many_bins_and_window_slides(backend, kind, nbins):
    """the sliding accumulator window of the conservative kernel (16 bins per lane in LDS): columns whose sweep
    runs forward, backward and back and forth over up to 64 bins, cells thicker than the window (read-modify-
    write beyond it), bins nothing ever reaches (NaN), more columns than one workgroup; 65 bins -> other kernel"""
    ncol, n = 300, 23
    if kind == "thick":
        rng = np.random.default_rng(11)
        theta = np.cumsum(rng.random((ncol, n + 1)) * 0.02, axis=-1)
        theta[:, 7:] += 3.0 + rng.random((ncol, 1))        # one cell spans most of the bins
        theta[::3, 15:] -= 2.5                              # and some columns come back down through them
    else:
        theta = _columns((ncol, n + 1), 13, kind)
    phi = R.synthetic_field((ncol, n), 14) * 100
    phi[5, 3] = np.nan
    top = float(np.nanmax(theta))
    bins = np.linspace(-0.3, top * 1.1 + 0.3, nbins + 1)
    bins[1:-1] += (R.synthetic_field((nbins - 1,), 15)) * (bins[1] - bins[0]) * 0.8   # uneven, still increasing
    bins[3] = theta[2, 4] if not np.isnan(theta[2, 4]) else bins[3]                   # an edge that coincides with a vertex
    bins = np.sort(bins)
    got = X.interp_1d_conservative(phi, theta, bins)
    want = TR.interp_1d_conservative(phi, theta, bins)
    assert got.shape == (ncol, nbins)
    np.testing.assert_array_equal(got, want)


def test_transform_on_a_zyx_field_without_transposes(backend):
    """(time, Z, Y, X) field, density-like target data: the column axis stays where it is in HBM;
    the result carries the reference's dim order (time, Y, X, target)."""
    nt, nz, ny, nx = 2, 9, 4, 6
    zc = np.arange(nz) + 0.5
    zo = np.arange(nz + 1) * 1.0
    ds = Dataset(coords={"Z": zc, "Zp1": zo, "Y": np.arange(ny), "X": np.arange(nx), "time": np.arange(nt)})
    grid = Grid(ds, coords={"Z": {"center": "Z", "outer": "Zp1"}}, autoparse_metadata=False)
    phi = R.synthetic_field((nt, nz, ny, nx), 21)
    dens = np.cumsum(R.synthetic_field((nt, nz, ny, nx), 22) + 0.6, axis=1)
    da = DataArray(phi, dims=("time", "Z", "Y", "X"), name="salt")
    sigma = DataArray(dens, dims=("time", "Z", "Y", "X"), name="sigma")
    levels = np.linspace(0.0, dens.max(), 7)
    out = grid.transform(da, "Z", levels, target_data=sigma)
    assert out.dims == ("time", "Y", "X", "sigma") and out.name == "salt"  # (`suffix` is never applied by the reference's `transform`: xgcm/transform.py:462-472)
    want = TR.interp_1d_linear(np.moveaxis(phi, 1, -1), np.moveaxis(dens, 1, -1), levels, mask_edges=True)
    np.testing.assert_array_equal(out.values, want)
    np.testing.assert_array_equal(out.coords["sigma"].values, levels)
    dens_o = np.cumsum(R.synthetic_field((nt, nz + 1, ny, nx), 23) + 0.6, axis=1)
    sig_o = DataArray(dens_o, dims=("time", "Zp1", "Y", "X"), name="sigma")
    out = grid.transform(da, "Z", levels, target_data=sig_o, method="conservative")
    assert out.dims == ("time", "Y", "X", "sigma")
    want = TR.interp_1d_conservative(np.moveaxis(phi, 1, -1), np.moveaxis(dens_o, 1, -1), levels)
    np.testing.assert_array_equal(out.values, want)
    np.testing.assert_array_equal(out.coords["sigma"].values, (levels[1:] + levels[:-1]) / 2)


@pytest.mark.gpu
def test_transform_resident_tensors():
    """HBM-resident phi / theta: the result is a resident tensor, a transposed VIEW in the reference's
    dim order (no copy), identical to the host path."""
    import torch

    from xgcm_amd import device as dev

    nz, ny, nx = 11, 6, 32
    ds = Dataset(coords={"Z": np.arange(nz) + 0.5, "Zp1": np.arange(nz + 1) * 1.0})
    grid = Grid(ds, coords={"Z": {"center": "Z", "outer": "Zp1"}}, autoparse_metadata=False)
    phi = R.synthetic_field((nz, ny, nx), 31)
    sig = np.cumsum(R.synthetic_field((nz, ny, nx), 32) + 0.6, axis=0)
    levels = np.linspace(0.2, sig.max(), 9)
    host = grid.transform(DataArray(phi, ("Z", "Y", "X"), name="s"), "Z", levels, target_data=DataArray(sig, ("Z", "Y", "X"), name="sigma"))
    res = grid.transform(DataArray(dev.asdevice(phi), ("Z", "Y", "X"), name="s"), "Z", levels,
                         target_data=DataArray(dev.asdevice(sig), ("Z", "Y", "X"), name="sigma"))
    assert isinstance(res.data, torch.Tensor) and res.data.is_cuda and res.dims == ("Y", "X", "sigma")
    assert not res.data.is_contiguous()  # (sigma, Y, X) in memory, presented as (Y, X, sigma)
    np.testing.assert_array_equal(res.values, host.values)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("lean", [0, 1])
def test_linear_transform_shared_targets_hard_cases(dtype, lean):
    """the lean streaming loop validates the shared targets once per workgroup: NaN / decreasing / repeated targets, targets
    on column points, left and right of every column, duplicated theta (zero-width intervals: numpy's NaN fall-backs),
    with and without edge masking and with the monotonicity checks bypassed -- the oracle's bits, as with the general loop"""
    from xgcm_amd import _hip
    from xgcm_amd import device as dev

    nz, ny, nx = 12, 2, 130
    rng = np.random.default_rng(5)
    theta = np.cumsum(rng.random((nz, ny, nx)) + 0.05, axis=0)
    theta[:, 0, 3] = theta[::-1, 0, 3]                 # decreasing column
    theta[5, 0, 10:14] = theta[4, 0, 10:14]            # duplicated theta: slope = x / 0
    theta[6, 1, 20] = np.nan
    theta[:, 1, 64:70] = np.round(theta[:, 1, 64:70])  # integers: targets fall exactly on points
    theta = theta.astype(dtype)
    phi = (R.synthetic_field((nz, ny, nx), 3) * 10).astype(dtype)
    phi[5, 0, 11] = np.inf                             # inf * 0 in the interpolation formula
    phi[4:6, 0, 12] = 7.0
    base = np.array([-1.0, 0.0, 1.0, 2.0, 2.0, 3.0, 4.5, 5.0, 6.0, 7.0, 50.0, 60.0])
    target_sets = {
        "sorted": base,
        "nan_inside": np.where(np.arange(base.size) == 5, np.nan, base),
        "nan_first": np.where(np.arange(base.size) == 0, np.nan, base),
        "decreasing": base[::-1].copy(),
        "one_inversion": np.concatenate([base[:4], [1.5], base[4:]]),
        "single": np.array([3.25]),
        "all_left": np.array([-5.0, -4.0]),
        "all_right": np.array([100.0, 101.0, 102.0]),
    }
    keep = _hip.get_tunable("transform_lean")
    _hip.set_tunable("transform_lean", 3 * lean)
    try:
        for name, tg in target_sets.items():
            tg = tg.astype(dtype)
            for mask_edges in (True, False):
                for bypass in (False, True):
                    want = np.moveaxis(TR.interp_1d_linear(np.moveaxis(phi, 0, -1), np.moveaxis(theta, 0, -1), tg,
                                                           mask_edges=mask_edges, bypass_checks=bypass), -1, 0)
                    got = dev.tohost(dev.transform_linear(phi, theta, tg.reshape(-1, 1, 1), 0, mask_edges=mask_edges,
                                                          bypass_checks=bypass))
                    np.testing.assert_array_equal(got, want, err_msg=f"{name} mask_edges={mask_edges} bypass={bypass}")
    finally:
        _hip.set_tunable("transform_lean", keep)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_linear_transform_output_staging_variants(dtype):
    """(Z, Y, X) layout, random-walk columns: the lanes of a wave emit a target level at different source levels.  Every
    way of getting the outputs to memory -- direct stores, level table in LDS, whole-column tile, ring of 4 / 8 / 16
    rows with lanes running ahead of the ring and columns that fall back to the exact search in the same wave --
    gives the oracle's bits."""
    from xgcm_amd import _hip
    from xgcm_amd import device as dev

    nz, ny, nx, m = 40, 3, 200, 50
    rng = np.random.default_rng(11)
    theta = np.cumsum(rng.random((nz, ny, nx)) * 2.0 + 0.01, axis=0)
    theta[:, 0, 5] = theta[::-1, 0, 5]          # a decreasing column (flipped)
    theta[7, 1, 70] = np.nan                     # NaN inside a column -> exact path
    theta[20:, 2, 130] = theta[19, 2, 130] - 1   # non-monotonic -> exact path
    theta[:, 1, 100:110] *= 0.05                 # compressed columns: all targets right of them, far ahead of the others
    theta[:, 2, 10:20] *= 30.0                   # stretched columns: lag behind
    theta = theta.astype(dtype)
    phi = (R.synthetic_field((nz, ny, nx), 12) * 10).astype(dtype)
    levels = np.linspace(0.5, 0.9 * float(np.nanmax(theta[:, 0, 0])), m).astype(dtype)
    want = np.moveaxis(TR.interp_1d_linear(np.moveaxis(phi, 0, -1), np.moveaxis(theta, 0, -1), levels, mask_edges=True), -1, 0)
    lv3 = levels.reshape(m, 1, 1)
    per_column = np.ascontiguousarray(np.broadcast_to(lv3, (m, ny, nx)))  # not shared: levels stay in memory
    keep = {k: _hip.get_tunable(k) for k in ("transform_stage", "transform_ring", "transform_win", "transform_cwin", "transform_lean")}
    try:
        for stage, ring, lean in ((0, 8, 0), (1, 8, 0), (2, 8, 0), (3, 4, 0), (3, 8, 0), (3, 16, 0), (3, 32, 0),
                                  (3, 4, 1), (3, 8, 1), (3, 16, 1), (3, 32, 1)):
            _hip.set_tunable("transform_stage", stage)
            _hip.set_tunable("transform_ring", ring)
            _hip.set_tunable("transform_lean", 3 * lean)
            for tg in (lv3, per_column):
                got = dev.tohost(dev.transform_linear(phi, theta, tg, 0, mask_edges=True))
                np.testing.assert_array_equal(got, want, err_msg=f"stage {stage} ring {ring} lean {lean}")
        # the conservative remap: one accumulator window per wave (4 / 8 / 16 bins, complete rows leave when every lane's
        # cursor has passed them) or per lane (transform_win=2), land columns and columns running backwards in the wave
        theta_o = np.concatenate([theta[:1] - 1.0, theta], axis=0)
        theta_o[:, 1, 30:40] = np.nan  # land
        edges = np.linspace(0.0, 1.1 * float(np.nanmax(theta[:, 0, 0])), m + 1).astype(dtype)
        want_c = np.moveaxis(TR.interp_1d_conservative(np.moveaxis(phi, 0, -1), np.moveaxis(theta_o, 0, -1), edges), -1, 0)
        for win, cwin, lean in ((2, 16, 0), (1, 4, 0), (1, 8, 0), (1, 16, 0), (1, 4, 3), (1, 8, 3), (1, 16, 3)):
            _hip.set_tunable("transform_win", win)
            _hip.set_tunable("transform_cwin", cwin)
            _hip.set_tunable("transform_lean", lean)  # bit 1: the lean one-window-per-wave kernel (K9e)
            got = dev.tohost(dev.transform_conservative(phi, theta_o, edges, 0))
            np.testing.assert_array_equal(got, want_c, err_msg=f"win {win} cwin {cwin} lean {lean}")
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


def test_transform_edge_shapes(backend):
    """empty batches, single-level columns, a single target level, targets all outside the column"""
    phi = R.synthetic_field((4, 6), 41)
    theta = np.cumsum(R.synthetic_field((4, 6), 42) + 0.6, axis=-1)
    assert X.interp_1d_linear(phi[:0], theta[:0], np.array([0.5, 1.0])).shape == (0, 2)
    assert X.interp_1d_conservative(phi[:0, :5], theta[:0], np.array([0.0, 1.0, 2.0])).shape == (0, 2)
    one = X.interp_1d_linear(phi[:, :1], theta[:, :1], np.array([-1.0, float(theta[0, 0]), 99.0]), mask_edges=False)
    np.testing.assert_array_equal(one, TR.interp_1d_linear(phi[:, :1], theta[:, :1], np.array([-1.0, float(theta[0, 0]), 99.0]), mask_edges=False))
    for lv in (np.array([1.5]), np.array([-5.0, -4.0]), np.array([1e3, 2e3])):
        for mask in (False, True):
            np.testing.assert_array_equal(X.interp_1d_linear(phi, theta, lv, mask_edges=mask),
                                          TR.interp_1d_linear(phi, theta, lv, mask_edges=mask))
    np.testing.assert_array_equal(X.interp_1d_conservative(phi[:, :5], theta, np.array([0.0, 100.0])),
                                  TR.interp_1d_conservative(phi[:, :5], theta, np.array([0.0, 100.0])))
    # 70 bins: beyond the LDS-accumulator budget of one kernel variant for float64 -> register-tile kernel
    bins = np.linspace(0.0, 6.0, 131)
    np.testing.assert_array_equal(X.interp_1d_conservative(phi[:, :5], theta, bins), TR.interp_1d_conservative(phi[:, :5], theta, bins))
