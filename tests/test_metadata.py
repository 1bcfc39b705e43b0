"""`Grid(ds)` from the dataset's own metadata: COMODO coordinate attributes and SGRID topology variables
(xgcm_amd/metadata.py; reference xgcm/metadata_parsers.py:4-45, comodo.py:23-142, sgrid.py:6-238 and their tests
xgcm/test/test_metadata_parsers.py, the COMODO datasets of xgcm/test/datasets.py:29-148 used throughout test_grid.py).
Datasets below are this suite's own (names, sizes); the expected position tables follow the conventions."""

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd import metadata as M

N = 9


def _comodo_1d(position, axis="X"):
    """a centre coordinate and one staggered coordinate at `position`"""
    n_other = {"left": N, "right": N, "inner": N - 1, "outer": N + 1}[position]
    # every staggered coordinate carries a shift; for inner / outer points the LENGTH decides, whatever its sign
    shift = {"left": -0.5, "right": 0.5, "inner": 0.5, "outer": -0.5}[position]
    start = {"left": 0.0, "right": 1.0, "inner": 1.0, "outer": 0.0}[position]
    attrs = {"axis": axis, "c_grid_axis_shift": shift}
    return Dataset({"tracer": (("t", "xc"), R.synthetic_field((3, N), 5))},
                   coords={"xc": ("xc", np.arange(N) + 0.5, {"axis": axis}),
                           "xs": ("xs", np.arange(n_other) + start, attrs), "t": ("t", np.arange(3.0))})


@pytest.mark.parametrize("position", ["left", "right", "inner", "outer"])
def test_comodo_positions_from_attributes_and_lengths(position):
    ds = _comodo_1d(position)
    assert M.parse_metadata(ds) == (ds, {"coords": {"X": {"center": "xc", position: "xs"}}})  # (ds, kwargs), xgcm/metadata_parsers.py:45
    assert M.parse_comodo(ds)[1]["coords"]["X"] == {"center": "xc", position: "xs"}
    assert list(M.parse_comodo(ds)[1]["coords"]["X"]) == ["center", position]  # centre first, like the reference's OrderedDict


def test_comodo_two_axes_and_unlabelled_dims():
    ds = Dataset(coords={"xc": ("xc", np.arange(6) + 0.5, {"axis": "X"}), "xg": ("xg", np.arange(6.0), {"axis": "X", "c_grid_axis_shift": -0.5}),
                         "yc": ("yc", np.arange(4) + 0.5, {"axis": "Y"}), "yp1": ("yp1", np.arange(5.0), {"axis": "Y", "c_grid_axis_shift": -0.5}),
                         "time": ("time", np.arange(3.0)), "k": ("k", np.arange(2.0), {"long_name": "no axis attribute"})})
    assert M.parse_metadata(ds)[1]["coords"] == {"X": {"center": "xc", "left": "xg"}, "Y": {"center": "yc", "outer": "yp1"}}
    assert M.parse_metadata(Dataset(coords={"time": ("time", np.arange(3.0))}))[1] == {"coords": {}}


def test_comodo_malformed_attributes():
    two = Dataset(coords={"a": ("a", np.arange(5.0), {"axis": "X"}), "b": ("b", np.arange(5.0), {"axis": "X"})})
    with pytest.raises(ValueError, match="Found two coordinates without `c_grid_axis_shift` attribute for axis X"):
        M.parse_metadata(two)
    none = Dataset(coords={"a": ("a", np.arange(5.0), {"axis": "X", "c_grid_axis_shift": -0.5})})
    with pytest.raises(ValueError, match="Couldn't find a center coordinate for axis X"):
        M.parse_metadata(none)
    short = Dataset(coords={"a": ("a", np.arange(5.0), {"axis": "X"}), "b": ("b", np.arange(3.0), {"axis": "X", "c_grid_axis_shift": -0.5})})
    with pytest.raises(ValueError, match=r"Left coordinate b has incompatible length 3 \(axis_len=5\)"):
        M.parse_metadata(short)
    odd = Dataset(coords={"a": ("a", np.arange(5.0), {"axis": "X"}), "b": ("b", np.arange(5.0), {"axis": "X", "c_grid_axis_shift": 0.25})})
    with pytest.raises(ValueError, match="Coordinate b has invalid `c_grid_axis_shift` attribute `0.25`"):
        M.parse_metadata(odd)
    # a shift that is set but not a number (old xmitgcm) still marks the coordinate as staggered: lengths decide
    listy = Dataset(coords={"a": ("a", np.arange(5.0), {"axis": "X"}), "b": ("b", np.arange(6.0), {"axis": "X", "c_grid_axis_shift": [-0.5]})})
    assert M.parse_metadata(listy)[1]["coords"]["X"] == {"center": "a", "outer": "b"}


def test_grid_from_comodo_metadata_computes(backend):
    """the autoparsed grid is the grid: same results as the one built from explicit coords"""
    ds = _comodo_1d("left")
    auto = Grid(ds, padding="periodic")
    explicit = Grid(ds, coords={"X": {"center": "xc", "left": "xs"}}, padding="periodic", autoparse_metadata=False)
    assert list(auto.axes) == ["X"] and auto.axes["X"].coords == explicit.axes["X"].coords
    for op in ("diff", "interp"):
        a, b = getattr(auto, op)(ds.tracer, "X"), getattr(explicit, op)(ds.tracer, "X")
        assert a.dims == b.dims == ("t", "xs")
        np.testing.assert_array_equal(a.values, b.values)
    np.testing.assert_array_equal(auto.diff(ds.tracer, "X").values, R.stencil1d("diff", ds.tracer.values, 1, 1, 0, "periodic"))


def test_autoparsed_and_explicit_kwargs_conflict_like_the_reference():
    """`coords` is always among the parsed kwargs (xgcm/grid.py:159-163): explicit coords need autoparse_metadata=False"""
    ds = _comodo_1d("left")
    with pytest.raises(ValueError, match="Autoparsed Grid kwargs: 'coords' conflict with user-supplied kwargs"):
        Grid(ds, coords={"X": {"center": "xc", "left": "xs"}})
    bare = Dataset({"v": (("x",), np.arange(4.0))})
    with pytest.raises(ValueError, match="Autoparsed Grid kwargs: 'coords' conflict"):
        Grid(bare, coords={"X": {"center": "x"}})
    assert dict(Grid(bare).axes) == {}  # nothing to parse is an empty grid there too (tests/test_metadata_reference.py)
    with pytest.raises(ValueError, match="Could not determine Axis names"):
        Grid(bare, autoparse_metadata=False)
    assert list(Grid(bare, coords={"X": {"center": "x"}}, autoparse_metadata=False).axes) == ["X"]


# ---- SGRID ------------------------------------------------------------------------------------------------------
def _sgrid(topology, conventions="SGRID-0.3", key="Conventions", dims=None):
    dims = dims or {"xi_rho": 7, "xi_psi": 6, "eta_rho": 5, "eta_psi": 4, "s_rho": 3, "s_w": 4}
    coords = {d: (d, np.arange(float(n))) for d, n in dims.items()}
    return Dataset({"topo": ((), np.array(1, dtype="int32"), dict(topology, cf_role="grid_topology"))}, coords=coords,
                   attrs={key: conventions})


SGRID_CASES = {
    "1d": ({"topology_dimension": 1, "node_dimensions": "xi_psi", "face_dimensions": "xi_rho: xi_psi (padding: both)"},
           {"X": {"center": "xi_rho", "inner": "xi_psi"}}),
    "2d": ({"topology_dimension": 2, "node_dimensions": "xi_psi eta_psi",
            "face_dimensions": "xi_rho: xi_psi (padding: both) eta_rho: eta_psi (padding: both)"},
           {"X": {"center": "xi_rho", "inner": "xi_psi"}, "Y": {"center": "eta_rho", "inner": "eta_psi"}}),
    "2d_no_space_after_colon": ({"topology_dimension": 2, "node_dimensions": "xi_psi eta_psi",
                                 "face_dimensions": "xi_rho:xi_psi (padding:high) eta_rho:eta_psi (padding:low)"},
                                {"X": {"center": "xi_rho", "left": "xi_psi"}, "Y": {"center": "eta_rho", "right": "eta_psi"}}),
    "2d_vertical": ({"topology_dimension": 2, "node_dimensions": "xi_psi eta_psi",
                     "face_dimensions": "xi_rho: xi_psi (padding: both) eta_rho: eta_psi (padding: both)",
                     "vertical_dimensions": "s_rho: s_w (padding: none)"},
                    {"X": {"center": "xi_rho", "inner": "xi_psi"}, "Y": {"center": "eta_rho", "inner": "eta_psi"},
                     "Z": {"center": "s_rho", "outer": "s_w"}}),
    "3d": ({"topology_dimension": 3, "node_dimensions": "xi_psi eta_psi s_w",
            "volume_dimensions": "xi_rho: xi_psi (padding: low) eta_rho: eta_psi (padding: high) s_rho: s_w (padding: none)"},
           {"X": {"center": "xi_rho", "right": "xi_psi"}, "Y": {"center": "eta_rho", "left": "eta_psi"}, "Z": {"center": "s_rho", "outer": "s_w"}}),
}


@pytest.mark.parametrize("case", SGRID_CASES)
def test_sgrid_topology_to_positions(case):
    topology, want = SGRID_CASES[case]
    for key, conv in (("Conventions", "SGRID-0.3"), ("conventions", "CF-1.8, sgrid-0.3")):
        ds = _sgrid(topology, conv, key)
        assert M.is_sgrid(ds)
        assert M.parse_metadata(ds)[1] == {"coords": want} == M.parse_sgrid(ds)[1]
    assert {ax: dict(a.coords) for ax, a in Grid(_sgrid(topology)).axes.items()} == want


def test_sgrid_refusals():
    assert not M.is_sgrid(_comodo_1d("left")) and not M.is_sgrid(_sgrid(SGRID_CASES["2d"][0], "CF-1.8"))
    with pytest.raises(ValueError, match="Could not find identify SGRID grid in input dataset."):
        M.parse_sgrid(_comodo_1d("left"))
    four = Dataset({"g": ((), np.array(1, dtype="int32"), {"cf_role": "grid_topology", "topology_dimension": 4})},
                   attrs={"Conventions": "SGRID-0.3"})
    with pytest.raises(ValueError, match="SGRID expected dataset with 1-3 spatial dimensions but got 4 in variable '.*'."):
        Grid(four)
    with pytest.raises(ValueError, match="'node_dimensions' attribute not found"):
        M.parse_metadata(_sgrid({"topology_dimension": 1, "face_dimensions": "xi_rho: xi_psi (padding: both)"}))
    with pytest.raises(IndexError, match="Not enough 'node_dimensions'"):
        M.parse_metadata(_sgrid({"topology_dimension": 2, "node_dimensions": "xi_psi",
                                 "face_dimensions": "xi_rho: xi_psi (padding: both) eta_rho: eta_psi (padding: both)"}))
    with pytest.raises(IndexError, match="Found 0 face_dimensions corresponding to node_dimension 'eta_psi'. Expecting 1."):
        M.parse_metadata(_sgrid({"topology_dimension": 2, "node_dimensions": "xi_psi eta_psi", "face_dimensions": "xi_rho: xi_psi (padding: both)"}))
    with pytest.raises(KeyError, match="Unexpected padding type 'sideways' in SGRID data."):
        M.parse_metadata(_sgrid({"topology_dimension": 1, "node_dimensions": "xi_psi", "face_dimensions": "xi_rho: xi_psi (padding: sideways)"}))


def test_grid_from_sgrid_metadata_computes(backend):
    topology, _ = SGRID_CASES["2d_vertical"]
    ds = _sgrid(topology)
    ds["temp"] = (("s_rho", "eta_rho", "xi_rho"), R.synthetic_field((3, 5, 7), 8))
    grid = Grid(ds, padding={"X": "extend", "Y": "extend", "Z": "fill"})
    out = grid.interp(ds.temp, "Z", to="outer")
    assert out.dims == ("s_w", "eta_rho", "xi_rho")
    np.testing.assert_array_equal(out.values, R.stencil1d("interp", ds.temp.values, 0, 1, 1, "fill"))
    assert grid.diff(ds.temp, "X").dims == ("s_rho", "eta_rho", "xi_psi")  # centre -> inner: one point fewer
