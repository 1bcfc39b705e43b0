"""`xgcm_amd.Grid` against the REFERENCE's own Grid stack, call for call.

tests/golden/grid_reference.{npz,json} were written by `oracle/make_golden_grid.py`: the reference's `xgcm/grid.py`,
`axis.py`, `grid_ufunc.py`, `padding.py`, `gridops.py`, `metrics.py` imported UNMODIFIED over `oracle/xr_min.py` (a
numpy-backed stand-in for the xarray calls they make; xarray itself is not installable in the image) and run on seeded
datasets: a line with all five positions (every position pair x diff / interp / min / max / cumsum x boundary modes x
reverse, defaults, misuse) and a C-grid box with metrics (one- and two-axis operators, `metric_weighted`, `derivative`,
`integrate`, `average`, `cumint`, `interp_like`, `get_metric` on fields with NaNs, extra coordinates and permuted dims).
Every call is replayed here on every backend and must give the same values, dims, name, coordinate names AND values --
or raise the same error.  Pinned MODULO THE STAND-IN: the control flow behind the fixtures is the
reference's; a difference between xr_min and real xarray is not caught here (tests/test_real_xarray.py is where real
xarray is met).

Bars: bit-exact for diff / interp / min / max / derivative / get_metric / interp_like; rtol 1e-12 for scans and reductions
(sums along the contiguous axis are re-associated by contract, multi-axis reductions run axis by axis).
"""

import json
import os
import warnings

import numpy as np
import pytest

from xgcm_amd import Dataset, Grid

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLDEN, "grid_reference.json")))
ARR = np.load(os.path.join(GOLDEN, "grid_reference.npz"))
EXACT = {"diff", "interp", "min", "max", "derivative", "get_metric", "interp_like"}
ERRORS = {"ValueError": ValueError, "KeyError": KeyError, "NotImplementedError": NotImplementedError, "TypeError": TypeError}


@pytest.fixture(params=["oracle-double", "host-abi", pytest.param("hip", marks=pytest.mark.gpu)])
def gbackend(request, monkeypatch):
    if request.param == "oracle-double":
        from oracle import fake_device

        fake_device.install(monkeypatch)
    elif request.param == "host-abi":
        import host_abi_device

        host_abi_device.install(monkeypatch)
    return request.param


# the user ufunc bodies of oracle/make_golden_grid.py::USER_UFUNCS (restated here: the fixtures hold data, not code)
def _second_order_diff(a):
    return a[..., 2:] - 2 * a[..., 1:-1] + a[..., :-2]


def _grad_inner(a):
    return a[..., 1:, 1:] - a[..., 1:, :-1], a[..., 1:, 1:] - a[..., :-1, 1:]


USER_UFUNCS = {
    "second_order_diff": (_second_order_diff, dict(axis=[("X",)], signature="(X:center)->(X:center)", padding_width={"X": (1, 1)})),
    "grad": (_grad_inner, dict(axis=[("Y", "X")], signature="(Y:center,X:center)->(Y:center,X:left),(Y:left,X:center)",
                               padding_width={"X": (1, 0), "Y": (1, 0)})),
    "cum_then_pad": (lambda a: np.cumsum(a, axis=-1)[..., :-1],
                     dict(axis=[("X",)], signature="(X:center)->(X:left)", padding_width={"X": (1, 0)}, pad_before_func=False,
                          padding="fill", fill_value=0.0)),
    "cum_untrimmed": (lambda a: np.cumsum(a, axis=-1),
                      dict(axis=[("X",)], signature="(X:center)->(X:left)", padding_width={"X": (1, 0)}, pad_before_func=False,
                           padding="fill", fill_value=0.0)),
}


def _links(fc):
    """face connections as `Grid` takes them: JSON turned the face numbers into strings and the links into lists"""
    out = {}
    for facedim, faces in fc.items():
        out[facedim] = {int(f): {ax: tuple(None if l is None else (int(l[0]), l[1], bool(l[2])) for l in pair) for ax, pair in axes.items()}
                        for f, axes in faces.items()}
    return out


def _build(name):
    d = META["datasets"][name]
    coords = {k: (tuple(v["dims"]), ARR[f"{name}/coord/{k}"], v["attrs"]) for k, v in d["coords"].items()}
    variables = {k: (tuple(v["dims"]), ARR[f"{name}/var/{k}"], v["attrs"]) for k, v in d["variables"].items()}
    ds = Dataset(variables, coords)
    gkw = dict(d["grid"])
    if "metrics" in gkw:
        gkw["metrics"] = {tuple(k.split("|")): v for k, v in gkw["metrics"].items()}
    if "face_connections" in gkw:
        gkw["face_connections"] = _links(gkw["face_connections"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        grid = Grid(ds, **gkw)
    return ds, grid


def _args(method, args, ds):
    out = []
    for a in args:
        if method == "interp_like" and isinstance(a, str):
            out.append(ds[a])
        elif method == "get_metric":
            out.append(tuple(a))
        else:
            out.append(a)
    return out


def _operand(spec, ds):
    if isinstance(spec, str) and spec.startswith("vec:"):
        _, ax, name = spec.split(":")
        return {ax: ds[name]}
    return ds[spec]


def _kwargs(kw, ds):
    out = {}
    for k, v in kw.items():
        if k == "metric_weighted" and isinstance(v, list):
            v = tuple(v)
        if k == "other_component":
            v = _operand(v, ds)
        out[k] = v
    return out


def _run(grid, ds, c):
    if c["method"].startswith("apply_as_grid_ufunc:"):
        func, ukw = USER_UFUNCS[c["method"].split(":")[1]]
        return grid.apply_as_grid_ufunc(func, _operand(c["var"], ds), **ukw)
    return getattr(grid, c["method"])(_operand(c["var"], ds), *_args(c["method"], c["args"], ds), **_kwargs(c["kwargs"], ds))


def _compare(res, key, exp, where, exact):
    want = ARR[key]
    assert list(res.dims) == exp["dims"], where
    assert res.name == exp["name"], f"{where}: name {res.name!r} vs {exp['name']!r}"
    assert sorted(res.coords) == exp["coords"], f"{where}: coords {sorted(res.coords)} vs {exp['coords']}"
    got = np.asarray(res.values)
    assert got.shape == want.shape and str(got.dtype) == exp["dtype"], where
    if exact:
        assert np.array_equal(got, want, equal_nan=True), f"{where}: max |diff| {np.nanmax(np.abs(got - want))}"
    else:
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=where)
    for cname in exp["coords"]:
        assert np.array_equal(np.asarray(res.coords[cname].values), ARR[f"{key}/coord/{cname}"]), f"{where}: coordinate {cname}"


CALLS_BY_DATASET = {name: [c for c in META["calls"] if c["dataset"] == name] for name in META["datasets"]}


@pytest.mark.parametrize("dsname", sorted(META["datasets"]))
def test_every_reference_call_replayed(gbackend, dsname):
    ds, grid = _build(dsname)
    topo = "face_connections" in META["datasets"][dsname]["grid"] or dsname == "fold"
    if topo and gbackend == "host-abi":
        pytest.skip("the host build of the ABI has no gather entry points (complex topologies)")
    checked = raised = 0
    for c in CALLS_BY_DATASET[dsname]:
        where = f"call {c['id']}: grid.{c['method']}({c['var']}, {c['args']}, {c['kwargs']})"
        if "raises" in c:
            with pytest.raises(ERRORS[c["raises"]["type"]]) as info:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    _run(grid, ds, c)
            got = str(info.value.args[0]) if info.value.args else str(info.value)
            want = c["raises"]["message"]
            if not want.startswith("all the input array dimensions"):   # (numpy's text from inside the stand-in's concat: type only)
                assert got[:60] == want[:60].strip("'\"") or got[:60] == want[:60], f"{where}: {got!r} vs {want!r}"
            raised += 1
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = _run(grid, ds, c)
        outs = list(res) if isinstance(res, (tuple, list)) else [res]
        exps = [c["result"]] + c.get("more_results", [])
        assert len(outs) == len(exps), where
        if c["method"] == "get_metric" and len(c["args"][0]) > 1 and sorted(outs[0].dims) == sorted(exps[0]["dims"]):
            # a PRODUCT of metrics takes its dim order from `itertools.combinations(frozenset(axes), k)` in the reference
            # (xgcm/metrics.py:22-24): the iteration order of a set of strings, i.e. of the process's hash seed -- not pinned
            outs[0] = outs[0].transpose(*exps[0]["dims"])
        exact = c["method"] in EXACT or (c["method"].startswith("apply_as_grid_ufunc") and "cum" not in c["method"])
        for k, (o, e) in enumerate(zip(outs, exps)):
            _compare(o, f"call/{c['id']}" if k == 0 else f"call/{c['id']}/out{k}", e, where, exact)
        checked += 1
    assert checked >= 12 and (raised >= 1 or dsname == "fold")
    assert checked + raised == len(CALLS_BY_DATASET[dsname])


def test_the_fixture_set_is_the_one_described():
    n_calls = len(META["calls"])
    assert n_calls >= 390 and sum("raises" in c for c in META["calls"]) >= 20
    assert set(META["datasets"]) == {"line", "box", "faces_x2x", "faces_x2y", "faces_x2y_rev", "cube", "fold"}
