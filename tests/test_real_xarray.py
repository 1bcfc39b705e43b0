"""The xarray surface against REAL xarray -- skipped wherever xarray is not importable (the build image, the GPU box).

Everything else in this suite meets xarray through tests/xarray_standin.py, and two pieces of xarray BEHAVIOUR are
restated rather than executed (oracle/refimpl.py header): the float default `skipna` of `DataArray.cumsum` / `.sum`
(reference xgcm/grid.py:1316,1605) and `DataArray.pad` keeping an integer dtype (xgcm/padding.py:610-615).  On any
box that has xarray these tests pin both, and the coordinate re-attachment (xgcm/grid_ufunc.py:1262-1320; reference
tests xgcm/test/test_grid.py:571-756), against xarray itself.  If the reference package is installed there as well
(`import xgcm`), the same calls are compared with it directly.  Nothing here reads /root/reference.
"""

import numpy as np
import pytest

# What is NOT pinned on the boxes of this build (no xarray / dask there) and the test below that pins it wherever they exist:
# the rows of DESIGN.md section 7's "Parity unpinned here" table, kept in step with it by tests/test_docs.py (which reads
# this literal without importing the module).
UNPINNED = {
    "skipna-cumsum": "test_cumsum_of_nan_data_is_xarrays_default_skipna",
    "skipna-sum": "test_integrate_and_average_of_nan_data_follow_xarray",
    "integer-pad": "test_integer_fields_follow_xarrays_pad_and_cumsum",
    "index-alignment": "test_arithmetic_on_differing_indexes_is_the_known_deviation",
    "set-order-corners": "test_corner_cells_of_a_two_axis_pad_against_the_installed_reference",
    "dask-chunks": "test_dask_chunked_input_equals_the_eager_result",
}

xr = pytest.importorskip("xarray")
if not hasattr(xr, "testing") or not hasattr(xr.DataArray, "cumsum"):  # tests/xarray_standin.py left in sys.modules
    pytest.skip("the module named xarray is not the real package", allow_module_level=True)

from xgcm_amd import Grid  # noqa: E402

N = 8


def _dataset(nan=False, dtype=np.float64):
    rng = np.random.default_rng(3)
    v = rng.standard_normal((N, N))
    if nan:
        v[2, 3] = np.nan
        v[5, 0] = np.nan
        v[7, 7] = np.nan
    if np.dtype(dtype).kind in "iu":
        v = rng.integers(-50, 50, (N, N)).astype(dtype)
    coords = {"XC": ("XC", np.arange(N) + 0.5), "XG": ("XG", np.arange(N) * 1.0), "time": ("time", np.arange(N) * 600.0),
              "t_label": ("time", np.arange(N).astype("int64")), "xc_aux": ("XC", np.arange(N).astype("int64") * 10),
              "lon_g": ("XG", np.arange(N) * 2.0, {"units": "degrees_east"})}
    return xr.Dataset({"v": (("time", "XC"), v.astype(dtype), {"units": "m s-1"}), "dx": (("XC",), rng.random(N) + 1.0)}, coords,
                      attrs={"title": "toy"})


def _grid(ds, **kw):
    kw.setdefault("padding", "periodic")
    return Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, metrics={("X",): ["dx"]}, autoparse_metadata=False, **kw)


@pytest.mark.parametrize("funcname", ["diff", "interp", "min", "max", "cumsum", "derivative", "cumint"])
def test_real_xarray_in_real_xarray_out(backend, funcname):
    """xgcm/test/test_grid.py:588-612 on real objects: dims, name, the coordinate set, coordinate attrs and dtypes"""
    ds = _dataset()
    out = getattr(_grid(ds), funcname)(ds["v"], "X")
    assert isinstance(out, xr.DataArray)
    # (`diff / dx`, `cumsum(v * dx)`: xarray keeps a name only while every operand carries it -- xgcm/grid.py:1576-1578, :1656-1660)
    assert out.dims == ("time", "XG") and out.name == (None if funcname in ("derivative", "cumint") else "v")
    assert set(out.coords) == {"time", "XG", "t_label", "lon_g"}  # xc_aux lives on the old core dim: gone
    xr.testing.assert_identical(out.coords["lon_g"].variable, ds["lon_g"].variable)  # values, dims, dtype AND attrs
    assert out.coords["lon_g"].attrs == {"units": "degrees_east"}
    assert out.coords["t_label"].dtype == np.int64
    np.testing.assert_array_equal(out["XG"].values, ds["XG"].values)


def test_recast_coords_on_noncore_dims_survive(backend):
    """xgcm/test/test_grid.py:647-703 (GH #496)"""
    ds = _dataset()
    new_time = (np.arange(N) * 600 / 3600.0).astype(np.float32)
    v = ds["v"].assign_coords(time=new_time, t_label=("time", (np.arange(N) + 100).astype(np.float32), {"long_name": "recast"}))
    grid = _grid(ds)
    for out in (grid.interp(v, "X"), grid.diff(v, "X"), grid.cumsum(v, "X", to="left")):
        assert out.coords["time"].dtype == np.float32 and out.coords["t_label"].dtype == np.float32
        np.testing.assert_array_equal(out.coords["time"].values, new_time)
        assert out.coords["t_label"].attrs == {"long_name": "recast"}
        assert "XC" not in out.dims and "xc_aux" not in out.coords


@pytest.mark.parametrize("to,reverse", [("left", False), ("left", True)])
@pytest.mark.parametrize("padding", ["fill", "extend", "periodic"])
def test_cumsum_of_nan_data_is_xarrays_default_skipna(backend, to, reverse, padding):
    """PINS `skipna`: the reference calls `da.cumsum(dim)` with xarray's float default (NaN counted as 0,
    xgcm/grid.py:1316); restated as numpy.nancumsum in oracle/refimpl.py:cumsum1d -- here against xarray itself,
    with the reference's trim / pad table (xgcm/grid.py:1326-1391) expressed in xarray operations."""
    ds = _dataset(nan=True)
    da = ds["v"]
    c = da.isel(XC=slice(None, None, -1)).cumsum("XC").isel(XC=slice(None, None, -1)) if reverse else da.cumsum("XC")
    mode = {"fill": "constant", "extend": "edge", "periodic": "wrap"}[padding]
    kw = {"constant_values": 0.0} if mode == "constant" else {}
    if not reverse:  # center -> left: drop last, pad (1, 0)
        want = c.isel(XC=slice(0, -1)).pad(XC=(1, 0), mode=mode, **kw)
    else:            # reversed center -> left: the cumulative sum itself
        want = c
    got = _grid(ds).cumsum(da, "X", to=to, padding=padding, fill_value=0.0, reverse=reverse)
    np.testing.assert_allclose(got.values, want.values, rtol=1e-12, atol=1e-12, equal_nan=True)
    assert not np.isnan(got.values).any()  # skipped, not propagated


def test_integrate_and_average_of_nan_data_follow_xarray(backend):
    """PINS `skipna` of `(da * metric).sum(dim)` (xgcm/grid.py:1605) and of `da.weighted(w).mean` (:1681-1685)"""
    ds = _dataset(nan=True)
    grid = _grid(ds)
    np.testing.assert_allclose(grid.integrate(ds["v"], "X").values, (ds["v"] * ds["dx"]).sum("XC").values, rtol=1e-12)
    np.testing.assert_allclose(grid.average(ds["v"], "X").values, ds["v"].weighted(ds["dx"]).mean("XC").values, rtol=1e-12)
    np.testing.assert_allclose(grid.integrate(ds["v"], "X", skipna=False).values,
                               (ds["v"] * ds["dx"]).sum("XC", skipna=False).values, rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("dtype", ["int16", "int64", "uint8"])
def test_integer_fields_follow_xarrays_pad_and_cumsum(backend, dtype):
    """PINS the integer assumptions of xgcm_amd.dtypes: `DataArray.pad(constant_values=...)` keeps the dtype and casts the
    constant (xgcm/padding.py:610-615), `DataArray.cumsum` of integers accumulates in int64 / uint64"""
    ds = _dataset(dtype=dtype)
    da = ds["v"]
    grid = _grid(ds, padding="fill", fill_value=3.7)
    padded = da.pad(XC=(1, 0), mode="constant", constant_values=3.7)
    want = padded.values[:, 1:] - padded.values[:, :-1]
    got = grid.diff(da, "X")
    assert got.dtype == want.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(got.values, want)
    c = grid.cumsum(da, "X", to="left", padding="fill", fill_value=0)
    wc = da.cumsum("XC").isel(XC=slice(0, -1)).pad(XC=(1, 0), mode="constant", constant_values=0)
    assert c.dtype == wc.dtype
    np.testing.assert_array_equal(c.values, wc.values)


def test_against_an_installed_reference_package(backend):
    """when the reference itself is installed next to xarray: the same calls, compared directly (values, dims, coords)"""
    xgcm = pytest.importorskip("xgcm")
    ds = _dataset(nan=True)
    ref = xgcm.Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, metrics={("X",): ["dx"]}, autoparse_metadata=False,
                    padding="periodic")
    mine = _grid(ds)
    for call in (lambda g: g.diff(ds["v"], "X"), lambda g: g.interp(ds["v"], "X"), lambda g: g.cumsum(ds["v"], "X", to="left"),
                 lambda g: g.derivative(ds["v"], "X"), lambda g: g.integrate(ds["v"], "X"), lambda g: g.cumint(ds["v"], "X", to="left")):
        a, b = call(mine), call(ref)
        assert a.dims == b.dims and set(a.coords) == set(b.coords)
        np.testing.assert_allclose(a.values, b.values, rtol=1e-12, atol=1e-12, equal_nan=True)


_CUBE = {"face": {0: {"X": ((3, "X", False), (1, "X", False)), "Y": ((4, "Y", False), (5, "Y", False))},
                  1: {"X": ((0, "X", False), (2, "X", False)), "Y": ((4, "X", False), (5, "X", True))},
                  2: {"X": ((1, "X", False), (3, "X", False)), "Y": ((4, "Y", True), (5, "Y", True))},
                  3: {"X": ((2, "X", False), (0, "X", False)), "Y": ((4, "X", True), (5, "X", False))},
                  4: {"X": ((3, "Y", True), (1, "Y", False)), "Y": ((2, "Y", True), (0, "Y", False))},
                  5: {"X": ((3, "Y", False), (1, "Y", True)), "Y": ((0, "Y", False), (2, "Y", True))}}}
_TWO = {"face": {0: {"X": (None, (1, "Y", True))}, 1: {"Y": (None, (0, "X", True))}}}


@pytest.mark.parametrize("conn", [_TWO, _CUBE], ids=["x2y_rev", "cubed_sphere"])
def test_face_connections_against_an_installed_reference_package(backend, conn):
    """PINS f2 without the stand-in: the reference's own `_pad_face_connections` on REAL xarray objects (the committed
    topology fixtures come from the same code over a numpy-backed stand-in for DataArray, oracle/make_golden_topology.py).
    Scalars and the two components of a C-grid vector, through `pad` and through the operators."""
    xgcm = pytest.importorskip("xgcm")
    from xgcm.padding import pad as ref_pad

    from xgcm_amd.padding import pad as my_pad

    nf, n = len(conn["face"]), 6
    rng = np.random.default_rng(23)
    ds = xr.Dataset({"c": (("face", "y", "x"), rng.standard_normal((nf, n, n))), "u": (("face", "y", "xl"), rng.standard_normal((nf, n, n))),
                     "v": (("face", "yl", "x"), rng.standard_normal((nf, n, n)))},
                    coords={"x": np.arange(n), "xl": np.arange(n) - 0.5, "y": np.arange(n), "yl": np.arange(n) - 0.5, "face": np.arange(nf)})
    coords = {"X": {"center": "x", "left": "xl"}, "Y": {"center": "y", "left": "yl"}}
    ref = xgcm.Grid(ds, coords=coords, face_connections=conn, autoparse_metadata=False)
    mine = Grid(ds, coords=coords, face_connections=conn, autoparse_metadata=False)
    def no_corners(arr, pw, ydim, xdim):
        """the reference walks the padded axes in `set` order (xgcm/padding.py:481-487: hash-seed dependent), and only the
        corner cells -- halo along both axes -- depend on it: compare everything else"""
        v = np.array(arr.transpose(..., ydim, xdim).values, dtype=np.float64)
        (yl, yh), (xl, xh) = pw.get("Y", (0, 0)), pw.get("X", (0, 0))
        ny, nx = v.shape[-2:]
        ycut = np.r_[0:yl, ny - yh:ny]
        xcut = np.r_[0:xl, nx - xh:nx]
        v[np.ix_(*[range(k) for k in v.shape[:-2]], ycut, xcut)] = 0.0
        return v

    for pw in ({"X": (1, 1)}, {"X": (2, 1), "Y": (1, 2)}):
        for mode in ("fill", "extend"):
            a = my_pad(ds["c"], mine, padding_width=dict(pw), padding=mode, fill_value=1.5)
            b = ref_pad(ds["c"], ref, padding_width=dict(pw), padding=mode, fill_value=1.5)
            assert a.dims == b.dims
            np.testing.assert_array_equal(no_corners(a, pw, "y", "x"), no_corners(b, pw, "y", "x"))
        for this, other in ((("X", "u"), ("Y", "v")), (("Y", "v"), ("X", "u"))):  # a vector component + its partner
            va = my_pad({this[0]: ds[this[1]]}, mine, padding_width=dict(pw), padding="fill", fill_value=0.0,
                        other_component={other[0]: ds[other[1]]})
            vb = ref_pad({this[0]: ds[this[1]]}, ref, padding_width=dict(pw), padding="fill", fill_value=0.0,
                         other_component={other[0]: ds[other[1]]})
            ydim = "yl" if this[1] == "v" else "y"
            xdim = "xl" if this[1] == "u" else "x"
            np.testing.assert_array_equal(no_corners(va, pw, ydim, xdim), no_corners(vb, pw, ydim, xdim))
    for op in ("diff", "interp"):
        for ax in ("X", "Y"):
            np.testing.assert_array_equal(getattr(mine, op)(ds["c"], ax, padding="extend").values,
                                          getattr(ref, op)(ds["c"], ax, padding="extend").values)


def test_grid_from_comodo_attributes_on_real_xarray(backend):
    """`Grid(ds)` with the reference's default `autoparse_metadata=True`: axes read off the coordinates' `axis` /
    `c_grid_axis_shift` attributes (xgcm/comodo.py:23-142), data variables untouched until a metric is asked for; the same
    grid as an installed reference package builds, where there is one"""
    ds = _dataset()
    ds["XC"].attrs["axis"] = "X"
    ds["XG"].attrs.update({"axis": "X", "c_grid_axis_shift": -0.5})
    grid = Grid(ds, padding="periodic", metrics={("X",): ["dx"]})
    assert dict(grid.axes["X"].coords) == {"center": "XC", "left": "XG"}
    with pytest.raises(ValueError, match="Autoparsed Grid kwargs: 'coords' conflict"):
        Grid(ds, coords={"X": {"center": "XC", "left": "XG"}})
    out = grid.derivative(ds["v"], "X")
    assert isinstance(out, xr.DataArray) and out.dims == ("time", "XG")
    try:
        import xgcm
    except ImportError:
        return
    ref = xgcm.Grid(ds, padding="periodic", metrics={("X",): ["dx"]})
    assert dict(ref.axes["X"].coords) == dict(grid.axes["X"].coords)
    np.testing.assert_allclose(out.values, ref.derivative(ds["v"], "X").values, rtol=1e-12, atol=1e-12)



def test_big_endian_variable_through_real_xarray(backend):
    """round 5: MDS / NetCDF-3 bytes as xarray holds them when a reader does not decode them (`np.fromfile(f, ">f4")`):
    numpy -- hence the reference -- computes them as they are and returns native arrays"""
    ds = _dataset()
    be = xr.DataArray(ds["v"].values.astype(">f4"), dims=("time", "XC"), name="v")
    grid = _grid(ds)
    for fn in ("diff", "interp", "cumsum"):
        out = getattr(grid, fn)(be, "X")
        assert isinstance(out, xr.DataArray) and out.dtype.isnative
        ref = getattr(grid, fn)(xr.DataArray(be.values.astype("<f4"), dims=("time", "XC"), name="v"), "X")
        np.testing.assert_array_equal(out.values, ref.values)
    want = np.roll(be.values, -0, axis=1) - np.roll(be.values, 1, axis=1)      # numpy on the big-endian array itself
    np.testing.assert_array_equal(grid.diff(be, "X").values, want)
    assert want.dtype == np.float32 and want.dtype.isnative


def test_deferred_results_with_real_xarray_operands(backend):
    """round 5: `Grid(..., fuse=True)` on xarray inputs -- deferred results, real xarray operands in the arithmetic,
    `.to_xarray()` at the end; equal to the eager chain on xarray objects"""
    from xgcm_amd import lazy

    ds = _dataset()
    eager = _grid(ds).diff(ds["v"], "X") / ds["dx"].rename({"XC": "XG"}).drop_vars(["xc_aux"], errors="ignore")
    q = _grid(ds, fuse=True).diff(ds["v"], "X") / ds["dx"].rename({"XC": "XG"})
    assert isinstance(q, lazy.LazyArray)
    back = q.to_xarray()
    assert isinstance(back, xr.DataArray) and back.dims == eager.dims
    np.testing.assert_array_equal(back.values, eager.values)


# ---- round 5, second half: what the reference's own suite and the differential fuzz asked for, on real xarray -------------
def test_private_state_the_reference_tests_read_is_real_xarray(backend):
    ds = _dataset()
    grid = _grid(ds)
    assert grid._ds is ds
    [m] = grid._metrics[frozenset(["X"])]
    assert isinstance(m, xr.DataArray) and m.dims == ("XC",)
    xr.testing.assert_equal(m, ds["dx"].reset_coords(drop=True))
    w = grid.get_metric(ds["v"], ("X",))
    assert isinstance(w, xr.DataArray)
    xr.testing.assert_allclose(grid.cumsum(ds["v"] * w, "X", padding="fill"), grid.cumint(ds["v"], "X", padding="fill"))
    assert isinstance(grid.interp_like(ds["dx"], grid.diff(ds["v"], "X")), xr.DataArray)


def test_direct_pad_on_real_xarray_keeps_attrs_and_strips_coords(backend):
    from xgcm_amd.padding import pad

    ds = _dataset()
    out = pad(ds["v"], _grid(ds), {"X": (2, 1)}, padding="periodic")
    want = ds["v"].reset_coords(drop=True).drop_vars(["time", "XC"]).pad(XC=(2, 1), mode="wrap")
    assert isinstance(out, xr.DataArray) and not list(out.coords)
    xr.testing.assert_identical(out, want)  # values, dims, name and attrs ({"units": "m s-1"}): DataArray.pad keeps attrs


def test_deferred_results_compute_to_real_xarray(backend):
    ds = _dataset()
    fused, eager = _grid(ds, fuse=True), _grid(ds)
    lazy = (fused.diff(ds["v"], "X") * 2.0) / ds["dx"].rename({"XC": "XG"}).assign_coords(XG=ds["XG"].values)
    out = lazy.compute()
    assert isinstance(out, xr.DataArray)
    want = (eager.diff(ds["v"], "X") * 2.0) / ds["dx"].rename({"XC": "XG"}).assign_coords(XG=ds["XG"].values)
    xr.testing.assert_allclose(out, want)
    assert out.dims == want.dims and out.name == want.name
    np.testing.assert_array_equal(np.asarray(fused.diff(ds["v"], "X")), eager.diff(ds["v"], "X").values)


def test_transform_keeps_the_input_name_like_the_reference(backend):
    """xgcm/transform.py:462-472: `suffix` is accepted and never applied"""
    nz, nx = 6, 4
    rng = np.random.default_rng(5)
    ds = xr.Dataset({"salt": (("x", "z"), rng.standard_normal((nx, nz))),
                     "sigma": (("x", "z"), np.cumsum(rng.random((nx, nz)) + 0.1, axis=1))},
                    {"z": ("z", np.arange(nz) + 0.5), "zo": ("zo", np.arange(nz + 1) * 1.0), "x": ("x", np.arange(nx) * 1.0)})
    grid = Grid(ds, coords={"Z": {"center": "z", "outer": "zo"}}, autoparse_metadata=False)
    out = grid.transform(ds["salt"], "Z", np.linspace(0.2, 3.0, 5), target_data=ds["sigma"])
    assert isinstance(out, xr.DataArray) and out.name == "salt" and out.dims == ("x", "sigma")
    try:
        import xgcm
    except ImportError:
        return
    ref = xgcm.Grid(ds, coords={"Z": {"center": "z", "outer": "zo"}}, autoparse_metadata=False)
    xr.testing.assert_identical(out, ref.transform(ds["salt"], "Z", np.linspace(0.2, 3.0, 5), target_data=ds["sigma"]))


def test_arithmetic_on_differing_indexes_is_the_known_deviation(backend):
    """`array * metric` / `array / metric` in the reference are xarray arithmetic (xgcm/grid.py:806-808,830-832): operands are
    ALIGNED on their index coordinates (inner join) first.  A field and the metrics of its own dataset share their indexes --
    pinned here: identical results -- and that is the case this backend serves; operands whose indexes DIFFER are the one
    xarray behaviour knowingly not reproduced: the left operand's labels are kept and nothing is dropped."""
    ds = _dataset()
    grid = _grid(ds)
    want = (ds["v"] * ds["dx"]).values
    got = grid.interp(ds["v"], "X", metric_weighted=("X",))  # multiplies by dx(XC) before, divides by the interpolated metric after
    assert isinstance(got, xr.DataArray) and got.shape == want.shape
    np.testing.assert_allclose(grid.integrate(ds["v"], "X").values, want.sum(axis=1), rtol=1e-12)
    shifted = ds["dx"].assign_coords(XC=ds["XC"].values + 1.0)  # labels 1.5 ... 8.5: xarray keeps the 7 common ones
    assert (ds["v"] * shifted).sizes["XC"] == N - 1
    from xgcm_amd.labeled import from_xarray

    ours = from_xarray(ds["v"]) * from_xarray(shifted)
    if ours.sizes["XC"] != N - 1:
        pytest.xfail("index alignment of arithmetic (inner join) is knowingly not reproduced: operands are taken to share their indexes")


@pytest.mark.parametrize("conn", ["x_to_x", "x_to_y"])
def test_corner_cells_of_a_two_axis_pad_against_the_installed_reference(backend, conn):
    """The reference pads several axes of a connected grid in the order of a `set` (xgcm/padding.py:481-487): only the CORNER
    cells of a two-axis pad depend on it, and their values follow PYTHONHASHSEED there.  Against an installed `xgcm`:
    everything but the corners must agree bit for bit; the corners agree whenever that process happened to walk the axes
    in the grid's order (reported, not asserted)."""
    xgcm = pytest.importorskip("xgcm")
    from xgcm.padding import pad as ref_pad

    from xgcm_amd.padding import pad as own_pad

    n = 6
    links = {"x_to_x": {"face": {0: {"X": (None, (1, "X", False))}, 1: {"X": ((0, "X", False), None)}}},
             "x_to_y": {"face": {0: {"X": (None, (1, "Y", False))}, 1: {"Y": ((0, "X", False), None)}}}}[conn]
    rng = np.random.default_rng(11)
    ds = xr.Dataset({"t": (("face", "y", "x"), rng.standard_normal((2, n, n)))},
                    {"x": np.arange(n) + 0.5, "xl": np.arange(n) * 1.0, "y": np.arange(n) + 0.5, "yl": np.arange(n) * 1.0, "face": [0, 1]})
    coords = {"X": {"center": "x", "left": "xl"}, "Y": {"center": "y", "left": "yl"}}
    ref = ref_pad(ds["t"], xgcm.Grid(ds, coords=coords, face_connections=links, padding="fill", autoparse_metadata=False),
                  padding_width={"X": (1, 1), "Y": (1, 1)}, padding="fill", fill_value=9.0)
    own = own_pad(ds["t"], Grid(ds, coords=coords, face_connections=links, padding="fill", autoparse_metadata=False),
                  padding_width={"X": (1, 1), "Y": (1, 1)}, padding="fill", fill_value=9.0)
    a, b = np.asarray(ref.transpose("face", "y", "x").values), np.asarray(own.transpose("face", "y", "x").values)
    assert a.shape == b.shape
    inner = np.ones(a.shape, dtype=bool)
    for j in (0, -1):
        for i in (0, -1):
            inner[:, j, i] = False
    np.testing.assert_array_equal(a[inner], b[inner])
    print("corners equal in this process:", bool(np.array_equal(a[~inner], b[~inner], equal_nan=True)))


def test_dask_chunked_input_equals_the_eager_result(backend):
    """A dask-backed variable is walked block by block (xgcm_amd.chunked; reference: `dask="parallelized"`, xgcm/grid.py:786-818)
    -- against real dask: same values as the eager call, a dask-backed result with the input's chunks, nothing computed on
    the way in; and the reference's refusal of inner / outer along a chunked core dim (xgcm/grid_ufunc.py:1136-1159)."""
    dsa = pytest.importorskip("dask.array")
    ds = _dataset()
    grid = _grid(ds)
    chunked = ds["v"].chunk({"time": 3})
    for call in (lambda v: grid.diff(v, "X"), lambda v: grid.cumsum(v, "X"), lambda v: grid.derivative(v, "X"), lambda v: grid.integrate(v, "X")):
        got, want = call(chunked), call(ds["v"])
        assert isinstance(got, xr.DataArray) and got.dims == want.dims
        np.testing.assert_array_equal(np.asarray(got.values), np.asarray(want.values))
    out = grid.diff(chunked, "X")
    assert isinstance(out.data, dsa.Array) and out.chunks[0] == chunked.chunks[0]
