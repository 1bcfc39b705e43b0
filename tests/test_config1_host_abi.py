"""BASELINE.json configs[0] -- `Grid.diff` along X on a 128 x 64 periodic 2-D C-grid, float64, "plumbing, no GPU" --
through the HOST build of the C ABI (libxgcm_host.so, g++): Grid -> dispatch -> fused grid ufunc -> device layer ->
ctypes -> the library's own loops, checked against the output of the reference's own ufunc bodies
(tests/golden/config1.npz) and, for the other 1-D operators, against the oracle.  The host build is for binding
tests; the product never loads it (see xgcm_amd/csrc/xg_host.cpp)."""

import ctypes
import os
import re

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid, _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_host_library_exports_the_whole_abi():
    header = open(os.path.join(ROOT, "include", "xgcm_hip.h")).read()
    declared = set(re.findall(r"^\s*int\s+(xg_\w+)\s*\(", header, flags=re.M))
    lib = ctypes.CDLL(os.path.join(ROOT, "xgcm_amd", "libxgcm_host.so"))
    assert all(hasattr(lib, n) for n in declared) and lib.xg_version() == 1 and lib.xg_device_count() == 0


def test_product_never_loads_the_host_library():
    """no fallback: the binding opens libxgcm_hip.so only, and without a GPU the device layer raises"""
    assert os.path.basename(_hip.LIB_PATH) == "libxgcm_hip.so" or os.environ.get("XG_HIP_LIB")
    src = "".join(open(os.path.join(ROOT, "xgcm_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "xgcm_amd")) if f.endswith(".py"))
    assert "libxgcm_host" not in src and "xg_host" not in src


def test_config1_through_the_host_abi(host_abi):
    fx = np.load(os.path.join(GOLDEN, "config1.npz"))
    from xgcm_amd import device as D  # (the product's own layer, its memory swapped for the host build by the fixture)

    T = D.tohost(D.synthetic((64, 128), 1))
    assert np.array_equal(T, fx["in"])  # the library's generator == the committed seed-1 input
    ds = Dataset({"T": (("YC", "XC"), T)}, coords={"XC": ("XC", np.arange(128) + 0.5), "XG": ("XG", np.arange(128) * 1.0),
                                                    "YC": ("YC", np.arange(64) * 1.0)})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, padding="periodic", autoparse_metadata=False)
    d = grid.diff(ds["T"], "X")
    assert d.dims == ("YC", "XG") and np.array_equal(d.values, fx["diff_X_center_to_left_periodic"])
    assert np.array_equal(grid.interp(ds["T"], "X").values, fx["interp_X_center_to_left_periodic"])


def test_the_other_1d_operators_through_the_host_abi(host_abi):
    nz, ny, nx = 4, 6, 10
    T = R.synthetic_field((nz, ny, nx), 3)
    T[1, 2, 3] = np.nan
    dx, drF = R.synthetic_metric((ny, nx), 4), R.synthetic_metric((nz,), 5)
    ds = Dataset({"dxC": (("YC", "XG"), dx), "drF": (("Z",), drF)},
                 coords={"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0, "YC": np.arange(ny) + 0.5, "YG": np.arange(ny) * 1.0,
                         "Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0, "Zp1": np.arange(nz + 1) * 1.0})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl", "outer": "Zp1"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, metrics={("X",): ["dxC"], ("Z",): ["drF"]},
                autoparse_metadata=False)
    da = DataArray(T, ("Z", "YC", "XC"))
    eq = lambda a, b: np.testing.assert_array_equal(a, b)  # noqa: E731
    eq(grid.min(da, "Y").values, R.stencil1d("min", T, 1, 1, 0, "extend"))
    eq(grid.max(da, "Z", to="outer").values, R.stencil1d("max", T, 0, 1, 1, "fill"))
    eq(grid.derivative(da, "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx[None]))
    eq(grid.cumsum(da, "Z", to="outer").values, R.grid_cumsum(T, 0, "center", "outer", "fill"))
    eq(grid.cumsum(da, "X").values, R.grid_cumsum(T, 2, "center", "left", "periodic"))
    eq(grid.integrate(da, "Z").values, R.integrate(T, 0, drF[:, None, None]))
    eq(grid.interp(da, ["X", "Y"]).values, R.stencil1d("interp", R.stencil1d("interp", T, 2, 1, 0, "periodic"), 1, 1, 0, "extend"))
    eq((da * 2.0 - da).values, T * 2.0 - T)
    # error path through the library: a message from xg_last_error, and the entry points outside the host build
    with pytest.raises(_hip.XgcmHipError, match="halo cells requested but no boundary mode"):
        __import__("xgcm_amd.device", fromlist=["x"]).stencil1d("diff", T, 2, 1, 0, None)
    with pytest.raises(_hip.XgcmHipError, match="not part of the host build"):
        grid.gradient(da)
    # (fused vorticity / divergence ARE in the host build since round 6 -- bench.py's config-5 leg in the CPU dry run of
    # tests/test_bench_dryrun.py -- and equal the operator chain bit for bit)
    u, v = DataArray(T, ("Z", "YC", "XG")), DataArray(R.synthetic_field((nz, ny, nx), 9), ("Z", "YG", "XC"))
    eq(grid.vorticity(u, v, metric_weighted=False).values, (grid.diff(v, "X") - grid.diff(u, "Y")).values)
    eq(grid.divergence(u, v, metric_weighted=False).values, (grid.diff(u, "X") + grid.diff(v, "Y")).values)


def test_host_build_knows_the_device_librarys_tunables():
    """same names in both builds (a binding written against one works against the other); unknown names are errors"""
    import re as _re

    names = _re.findall(r'\{"(\w+)", &Tune::', open(os.path.join(ROOT, "xgcm_amd", "csrc", "xg_runtime.hip")).read())
    lib = ctypes.CDLL(os.path.join(ROOT, "xgcm_amd", "libxgcm_host.so"))
    v = ctypes.c_int(-1)
    for n in names:
        assert lib.xg_set_tunable(n.encode(), 7) == 0 and lib.xg_get_tunable(n.encode(), ctypes.byref(v)) == 0 and v.value == 7
        assert _hip.get_tunable(n) is not None  # and the device library knows it too
    assert lib.xg_set_tunable(b"no_such_knob", 1) != 0
