"""More of the reference's own Grid / metrics tests, mirrored on seeded data (file:line each).

Complements tests/test_grid_api.py: metric bookkeeping (`set_metrics`, overwrite rules, error
messages), `interp_like`, constructor and input-validation errors of the grid methods and of
`apply_as_grid_ufunc`.  Runs on CPU (oracle-backed device double) and, marked gpu, through the C ABI.
"""

import numpy as np
import pytest

from xgcm_amd import DataArray, Dataset, Grid

from test_grid_api import _np, cgrid


def _drop(da):
    return da.reset_coords(drop=True)


def test_set_metric(backend):
    """test_metrics.py:329-351."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, autoparse_metadata=False)
    manual = Grid(ds, coords=coords, autoparse_metadata=False)
    for key, value in metrics.items():
        manual.set_metrics(key, value)
    assert len(grid._metrics) > 0
    for k, names in metrics.items():
        k = frozenset(k)
        assert k in grid._metrics and k in manual._metrics
        for name, m1, m2 in zip(names, grid._metrics[k], manual._metrics[k]):
            want = _drop(ds[name])
            assert m1.dims == want.dims and m2.dims == want.dims and all(c in m1.dims for c in m1.coords)
            np.testing.assert_array_equal(_np(m1), _np(want))
            np.testing.assert_array_equal(_np(m2), _np(want))


@pytest.mark.parametrize(
    "metric_axes, existing, add, expected",
    [
        ("X", ["dx_t", "dx_n", "dx_e", "dx_ne"], ["dx_n_overwrite"], ["dx_t", "dx_n_overwrite", "dx_e", "dx_ne"]),
        (("Y", "X"), ["area_t", "area_n", "area_e", "area_ne"], ["area_n_overwrite"],
         ["area_t", "area_n_overwrite", "area_e", "area_ne"]),
        ("X", ["dx_t", "dx_n", "dx_e"], ["dx_n_overwrite", "dx_ne"], ["dx_t", "dx_n_overwrite", "dx_e", "dx_ne"]),
    ],
)
def test_set_metric_overwrite_true(backend, metric_axes, existing, add, expected):
    """test_metrics.py:354-404: overwrite replaces the metric with the same dims, new dims append."""
    ds, coords, metrics = cgrid()
    ds[add[0]] = _drop(ds[existing[1]]) * 10
    metrics = {k: [m for m in v if m in existing] for k, v in metrics.items()}
    grid = Grid(ds, coords=coords, metrics=metrics, autoparse_metadata=False)
    for name in add:
        grid.set_metrics(metric_axes, name, overwrite=True)
    got = grid._metrics[frozenset(list(metric_axes))]
    assert len(got) == len(expected)
    for m, name in zip(got, expected):
        want = _drop(ds[name])
        assert m.dims == want.dims
        np.testing.assert_array_equal(_np(m), _np(want))


@pytest.mark.parametrize("metric_axes, overwrite_metric, add_metric", [("X", "dx_t_overwrite", "dx_t"), ("X", "dx_e", None)])
def test_set_metric_value_errors(backend, metric_axes, overwrite_metric, add_metric):
    """test_metrics.py:407-420."""
    ds, coords, metrics = cgrid()
    if add_metric is not None:
        ds[overwrite_metric] = _drop(ds[add_metric]) * 10
    grid = Grid(ds, coords=coords, metrics=metrics, autoparse_metadata=False)
    with pytest.raises(ValueError, match="setting overwrite=True."):
        grid.set_metrics(metric_axes, overwrite_metric)


def test_set_metric_key_errors(backend):
    """test_metrics.py:423-436."""
    ds, coords, metrics = cgrid()
    grid = Grid(ds, coords=coords, metrics=metrics, autoparse_metadata=False)
    with pytest.raises(KeyError, match="not found in dataset."):
        grid.set_metrics("X", "foo")
    with pytest.raises(KeyError, match="not compatible with grid axes"):
        grid.set_metrics(("U", "V"), "area_n")


@pytest.mark.parametrize("metric_axes, metric_name", [(["Y", "X"], "area_n"), ("X", "dx_t"), ("Y", "dy_ne"),
                                                       (["Y", "X"], "dy_n"), (["X"], "tracer")])
@pytest.mark.parametrize("grid_padding", ["periodic", "fill"])
@pytest.mark.parametrize("padding, padding_expected", [
    ({"X": "fill", "Y": "fill"}, {"X": "fill", "Y": "fill"}), ({"X": "extend", "Y": "extend"}, {"X": "extend", "Y": "extend"}),
    ("fill", {"X": "fill", "Y": "fill"}), ("extend", {"X": "extend", "Y": "extend"}),
    ({"X": "extend", "Y": "fill"}, {"X": "extend", "Y": "fill"})])
@pytest.mark.parametrize("fill_value", [None, 0.1])
def test_interp_like(backend, metric_axes, metric_name, grid_padding, padding, padding_expected, fill_value):
    """test_grid.py:776-819: `interp_like(array, like)` == interp along the axes where positions differ."""
    ds, coords, _ = cgrid()
    grid = Grid(ds, coords=coords, padding=grid_padding, autoparse_metadata=False)
    grid.set_metrics(metric_axes, metric_name)
    available = grid._metrics[frozenset(metric_axes)][0]
    got = grid.interp_like(available, ds["u"], padding=padding, fill_value=fill_value)
    # the reference interpolates along `metric_axes`; positions only differ from `u` on a subset of
    # them, and interp along an axis where they agree would move the array away: compare on that subset
    differ = [ax for ax in ("X", "Y", "Z") if ax in grid.axes and any(d in available.dims for d in grid.axes[ax].coords.values())
              and grid.axes[ax]._get_position_name(available)[0] != grid.axes[ax]._get_position_name(ds["u"])[0]]
    want = grid.interp(_drop(ds[metric_name]), differ, padding=padding_expected, fill_value=fill_value) if differ else _drop(ds[metric_name])
    assert got.dims == want.dims
    np.testing.assert_array_equal(_np(got), _np(want))
    with pytest.raises(ValueError, match="renamed to 'padding'"):
        grid.interp_like(available, ds["u"], boundary="fill")


def test_constructor_dim_errors(backend):
    """test_grid.py:822-840."""
    ds = Dataset({"data": (("x", "y"), np.zeros((4, 5)))}, coords={"c": (("x", "y"), np.ones((4, 5)))})
    with pytest.raises(ValueError, match="Could not find dimension"):
        Grid(ds, coords={"X": {"center": "c"}}, autoparse_metadata=False)
    msg = r"Could not find dimension `other` \(for the `center` position on axis `X`\) in input dataset."
    with pytest.raises(ValueError, match=msg):
        Grid(ds, coords={"X": {"center": "other"}}, autoparse_metadata=False)


def test_input_errors_of_grid_methods_and_apply(backend):
    """test_grid.py:895-1020 (TestInputErrorGridMethods / TestInputErrorApplyAsGridUfunc)."""
    ds, coords, _ = cgrid()
    grid = Grid(ds, coords=coords, autoparse_metadata=False)
    empty = DataArray(np.zeros(()), dims=())
    cases = [
        (ValueError, "Vector components provided as dictionaries should contain exactly one key/value pair.",
         {"X": empty, "Y": empty}),
        (TypeError, "All data arguments must be either a DataArray or Dictionary", "not_a_dataarray"),
        (TypeError, "Dictionary inputs must have a DataArray as value. Got", {"X": "not_a_dataarray"}),
        (ValueError, "Vector component with unknown axis provided. Grid has axes", {"wrong": empty}),
    ]
    for exc, msg, arg in cases:
        with pytest.raises(exc, match=msg):
            grid.diff(arg, "X")
        with pytest.raises(exc, match=msg):
            grid.apply_as_grid_ufunc(lambda x: x, arg, axis=[("X",)], signature="(X:center)->(X:center)")
    with pytest.raises(ValueError, match="When providing multiple input arguments, `other_component` needs to provide one dictionary per input"):
        grid.apply_as_grid_ufunc(lambda x: x, {"X": empty}, {"Y": empty}, {"Z": empty}, axis="X",
                                 other_component=[{"X": empty}, {"Y": empty}])


def test_vector_dict_input_without_face_connections(backend):
    """test_grid.py `test_2d_vector_dict_input_no_face_connections`: dict inputs behave like the bare array."""
    ds, coords, _ = cgrid()
    grid = Grid(ds, coords=coords, padding="fill", autoparse_metadata=False)
    a = grid.interp({"X": ds["u"]}, "X", other_component={"Y": ds["v"]})
    b = grid.interp(ds["u"], "X")
    assert a.dims == b.dims
    np.testing.assert_array_equal(_np(a), _np(b))


def test_default_boundary_is_not_periodic_and_bad_values(backend):
    """test_grid.py:`test_default_boundary_is_not_periodic`, `test_invalid_boundary_error`, `test_invalid_fill_value_error`."""
    ds, coords, _ = cgrid()
    grid = Grid(ds, coords=coords, autoparse_metadata=False)
    assert all(ax.padding is None and not ax.periodic for ax in grid.axes.values())
    with pytest.raises(ValueError, match="No boundary condition was specified"):
        grid.diff(ds["tracer"], "X")
    with pytest.raises(ValueError, match="padding must be one of"):
        Grid(ds, coords=coords, padding="bad", autoparse_metadata=False)
    with pytest.raises(TypeError, match="fill value must be an integer or a float"):
        Grid(ds, coords=coords, fill_value="x", autoparse_metadata=False)
