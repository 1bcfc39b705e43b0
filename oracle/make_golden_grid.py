#!/usr/bin/env python3
"""Golden fixtures for the `Grid` LEVEL of the hot path from the REFERENCE's own Grid stack.

TEST INFRASTRUCTURE -- build container only (reads /root/reference), never shipped, never on the GPU box.

Until round 5 everything above the raw ufunc bodies (dispatch, per-axis kwargs, the cumsum trim / pad table, metric
selection and interpolation, derivative / integrate / average / cumint, coordinate re-attachment, dim order) was RESTATED
from reading `xgcm/grid.py`, `grid_ufunc.py`, `axis.py`, `padding.py` and pinned by transcribed known answers only: xarray
is not installable here, so `import xgcm` fails.  This script registers `oracle/xr_min.py` -- a numpy-backed stand-in
for the few dozen xarray calls those files make -- under the name `xarray`, imports the reference's modules UNMODIFIED,
builds seeded datasets, runs the reference's own `Grid` methods on them and records inputs, outputs (values, dims, name,
coordinate names and values) and raised errors:

    tests/golden/grid_reference.npz   arrays (inputs + expected outputs)
    tests/golden/grid_reference.json  datasets, grids, the list of calls

`tests/test_grid_reference.py` replays the calls through `xgcm_amd.Grid` on every backend.  PINNED MODULO THE STAND-IN: the
control flow that produced the fixtures is the reference's, line for line; the container semantics are xr_min's.

    python oracle/make_golden_grid.py
"""
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    from oracle import xr_min

    xr = types.ModuleType("xarray")
    for name in ("DataArray", "Dataset", "apply_ufunc", "concat"):
        setattr(xr, name, getattr(xr_min, name))
    sys.modules["xarray"] = xr
    dask = types.ModuleType("dask")
    dask_array = types.ModuleType("dask.array")
    dask_array.Array = type("Array", (), {})
    dask.array = dask_array
    sys.modules["dask"] = dask
    sys.modules["dask.array"] = dask_array
    sys.path.insert(0, REF)
    import xgcm.grid as grid  # noqa

    return xr_min, grid


def field(rng, shape, nan_at=()):
    a = rng.standard_normal(shape)
    flat = a.reshape(-1)
    for k in nan_at:
        flat[k % flat.size] = np.nan
    return a


def datasets(rng):
    """name -> (variables {name: (dims, values[, attrs])}, coords, grid kwargs)"""
    out = {}
    # ---- one axis with every position ------------------------------------------------------------------------------
    n = 9
    coords = {"xc": ("xc", np.arange(n) + 0.5), "xl": ("xl", np.arange(n) * 1.0), "xr": ("xr", np.arange(n) + 1.0),
              "xi": ("xi", np.arange(1, n) * 1.0), "xo": ("xo", np.arange(n + 1) * 1.0), "t": ("t", np.arange(3) * 10.0),
              "label": ("t", np.array([7, 8, 9])), "lon": ("xc", np.linspace(0, 80, n), {"units": "degrees_east"})}
    variables = {f"f_{p}": (("t", d), field(rng, (3, m), nan_at=(4,) if p == "center" else ()))
                 for p, d, m in (("center", "xc", n), ("left", "xl", n), ("right", "xr", n), ("inner", "xi", n - 1), ("outer", "xo", n + 1))}
    variables["dx_c"] = (("xc",), rng.random(n) + 0.5)
    variables["dx_l"] = (("xl",), rng.random(n) + 0.5)
    variables["dx_o"] = (("xo",), rng.random(n + 1) + 0.5)
    out["line"] = (variables, coords,
                   {"coords": {"X": {"center": "xc", "left": "xl", "right": "xr", "inner": "xi", "outer": "xo"}},
                    "metrics": {("X",): ["dx_c", "dx_l", "dx_o"]}, "autoparse_metadata": False})
    # ---- a C-grid box with metrics ------------------------------------------------------------------------------------
    nz, ny, nx = 4, 5, 6
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", -np.arange(nz) - 0.5), "Zl": ("Zl", -np.arange(nz) * 1.0),
              "Zp1": ("Zp1", -np.arange(nz + 1) * 1.0), "time": ("time", np.array([0.0, 3600.0])),
              "iter": ("time", np.array([0, 72])), "depth": (("YC", "XC"), rng.random((ny, nx)) * 100.0)}
    variables = {
        "T": (("time", "Z", "YC", "XC"), field(rng, (2, nz, ny, nx), nan_at=(17, 101)), {"units": "degC"}),
        "U": (("time", "Z", "YC", "XG"), field(rng, (2, nz, ny, nx))),
        "V": (("time", "Z", "YG", "XC"), field(rng, (2, nz, ny, nx))),
        "W": (("time", "Zl", "YC", "XC"), field(rng, (2, nz, ny, nx))),
        "Tyx": (("YC", "XC"), field(rng, (ny, nx))),
        "Txyz": (("XC", "YC", "Z"), field(rng, (nx, ny, nz))),          # dims in another order: output order must follow
        "dxC": (("YC", "XG"), rng.random((ny, nx)) + 0.5), "dxG": (("YG", "XC"), rng.random((ny, nx)) + 0.5),
        "dxF": (("YC", "XC"), rng.random((ny, nx)) + 0.5), "dxV": (("YG", "XG"), rng.random((ny, nx)) + 0.5),
        "dyC": (("YG", "XC"), rng.random((ny, nx)) + 0.5), "dyG": (("YC", "XG"), rng.random((ny, nx)) + 0.5),
        "dyF": (("YC", "XC"), rng.random((ny, nx)) + 0.5), "dyU": (("YG", "XG"), rng.random((ny, nx)) + 0.5),
        "drF": (("Z",), rng.random(nz) + 0.5), "drC": (("Zp1",), rng.random(nz + 1) + 0.5),
        "rA": (("YC", "XC"), rng.random((ny, nx)) + 0.5), "rAz": (("YG", "XG"), rng.random((ny, nx)) + 0.5),
        "hFacC": (("Z", "YC", "XC"), rng.random((nz, ny, nx)) + 0.5),
    }
    out["box"] = (variables, coords,
                  {"coords": {"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                              "Z": {"center": "Z", "left": "Zl", "outer": "Zp1"}},
                   "padding": {"X": "periodic", "Y": "extend", "Z": "fill"}, "fill_value": {"Z": 0.0},
                   "metrics": {("X",): ["dxC", "dxG", "dxF", "dxV"], ("Y",): ["dyC", "dyG", "dyF", "dyU"],
                               ("Z",): ["drF", "drC"], ("X", "Y"): ["rA", "rAz"]},
                   "autoparse_metadata": False})
    # ---- connected faces (two faces; a cubed sphere) and a tripolar fold: the Grid over the reference's topology code ------------
    n = 5
    fcoords = {"x": ("x", np.arange(n) + 0.5), "xl": ("xl", np.arange(n) * 1.0), "y": ("y", np.arange(n) + 0.5), "yl": ("yl", np.arange(n) * 1.0)}
    axes = {"X": {"center": "x", "left": "xl"}, "Y": {"center": "y", "left": "yl"}}
    for name, nf, conn in (("faces_x2x", 2, X_TO_X), ("faces_x2y", 2, X_TO_Y), ("faces_x2y_rev", 2, X_TO_Y_REV), ("cube", 6, CUBED_SPHERE)):
        variables = {"c": (("z", "face", "y", "x"), field(rng, (2, nf, n, n))), "u": (("z", "face", "y", "xl"), field(rng, (2, nf, n, n))),
                     "v": (("z", "face", "yl", "x"), field(rng, (2, nf, n, n)))}
        coords = dict(fcoords, face=("face", np.arange(nf)), z=("z", np.arange(2) * 1.0))
        out[name] = (variables, coords, {"coords": axes, "face_connections": conn, "padding": "fill", "fill_value": 1.5,
                                         "autoparse_metadata": False})
    ny, nx = 6, 8
    variables = {"t": (("z", "yh", "xh"), field(rng, (2, ny, nx))), "u": (("z", "yh", "xq"), field(rng, (2, ny, nx))),
                 "v": (("z", "yq", "xh"), field(rng, (2, ny, nx)))}
    coords = {"xh": ("xh", np.arange(nx) + 0.5), "xq": ("xq", np.arange(nx) + 1.0), "yh": ("yh", np.arange(ny) + 0.5),
              "yq": ("yq", np.arange(ny) + 1.0), "z": ("z", np.arange(2) * 1.0)}
    out["fold"] = (variables, coords, {"coords": {"X": {"center": "xh", "right": "xq"}, "Y": {"center": "yh", "right": "yq"}},
                                       "padding": {"X": "periodic", "Y": {"fold": "corner", "south": "extend"}}, "autoparse_metadata": False})
    return out


X_TO_X = {"face": {0: {"X": (None, (1, "X", False))}, 1: {"X": ((0, "X", False), None)}}}
X_TO_Y = {"face": {0: {"X": (None, (1, "Y", False))}, 1: {"Y": ((0, "X", False), None)}}}
X_TO_Y_REV = {"face": {0: {"X": (None, (1, "Y", True))}, 1: {"Y": (None, (0, "X", True))}}}
CUBED_SPHERE = {
    "face": {
        0: {"X": ((3, "X", False), (1, "X", False)), "Y": ((4, "Y", False), (5, "Y", False))},
        1: {"X": ((0, "X", False), (2, "X", False)), "Y": ((4, "X", False), (5, "X", True))},
        2: {"X": ((1, "X", False), (3, "X", False)), "Y": ((4, "Y", True), (5, "Y", True))},
        3: {"X": ((2, "X", False), (0, "X", False)), "Y": ((4, "X", True), (5, "X", False))},
        4: {"X": ((3, "Y", True), (1, "Y", False)), "Y": ((2, "Y", True), (0, "Y", False))},
        5: {"X": ((3, "Y", False), (1, "Y", True)), "Y": ((0, "Y", False), (2, "Y", True))},
    }
}


# user ufuncs (xgcm/grid_ufunc.py:661-951 `apply_as_grid_ufunc`): the SAME numpy bodies are in tests/test_grid_reference.py
def _second_order_diff(a):
    return a[..., 2:] - 2 * a[..., 1:-1] + a[..., :-2]


def _grad_inner(a):
    return a[..., 1:, 1:] - a[..., 1:, :-1], a[..., 1:, 1:] - a[..., :-1, 1:]


def _cum_then_pad(a):
    return np.cumsum(a, axis=-1)[..., :-1]


def _cum_untrimmed(a):  # one cell too long for its padding: the reference's "does your grid ufunc correctly trim" error
    return np.cumsum(a, axis=-1)


USER_UFUNCS = {
    "second_order_diff": (_second_order_diff, dict(axis=[("X",)], signature="(X:center)->(X:center)", padding_width={"X": (1, 1)})),
    "grad": (_grad_inner, dict(axis=[("Y", "X")], signature="(Y:center,X:center)->(Y:center,X:left),(Y:left,X:center)",
                               padding_width={"X": (1, 0), "Y": (1, 0)})),
    "cum_then_pad": (_cum_then_pad, dict(axis=[("X",)], signature="(X:center)->(X:left)", padding_width={"X": (1, 0)},
                                         pad_before_func=False, padding="fill", fill_value=0.0)),
    "cum_untrimmed": (_cum_untrimmed, dict(axis=[("X",)], signature="(X:center)->(X:left)", padding_width={"X": (1, 0)},
                                           pad_before_func=False, padding="fill", fill_value=0.0)),
}


def calls():
    """(dataset, method, variable, positional args, kwargs)"""
    c = []
    positions = {"center": ["left", "right", "inner", "outer"], "left": ["center"], "right": ["center"], "inner": ["center"],
                 "outer": ["center"]}
    for pos, targets in positions.items():
        for to in targets:
            for op in ("diff", "interp", "min", "max"):
                for padding, fill in (("periodic", None), ("extend", None), ("fill", 2.5)):
                    kw = {"to": to, "padding": padding}
                    if fill is not None:
                        kw["fill_value"] = fill
                    c.append(("line", op, f"f_{pos}", ["X"], kw))
            for padding, fill in (("fill", None), ("fill", 1.5), ("extend", None), ("periodic", None), (None, None)):
                for reverse in (False, True):
                    kw = {"to": to, "reverse": reverse}
                    if padding is not None:
                        kw["padding"] = padding
                    if fill is not None:
                        kw["fill_value"] = fill
                    c.append(("line", "cumsum", f"f_{pos}", ["X"], kw))
        c.append(("line", "diff", f"f_{pos}", ["X"], {"padding": "periodic"}))      # default shift
        c.append(("line", "interp", f"f_{pos}", ["X"], {}))                          # no boundary given: may raise
    c += [("line", "derivative", "f_center", ["X"], {"padding": "extend"}), ("line", "derivative", "f_left", ["X"], {"padding": "periodic"}),
          ("line", "integrate", "f_center", ["X"], {}), ("line", "average", "f_center", ["X"], {}),
          ("line", "cumint", "f_center", ["X"], {"padding": "fill"}), ("line", "cumint", "f_center", ["X"], {"to": "outer", "padding": "fill"}),
          ("line", "interp", "f_center", ["X"], {"to": "outer", "padding": "extend", "metric_weighted": "X"}),
          ("line", "diff", "f_center", ["Y"], {}), ("line", "diff", "f_center", ["X"], {"to": "centre", "padding": "fill"}),
          ("line", "cumsum", "f_left", ["X"], {"to": "right", "padding": "fill"})]
    for var in ("T", "Tyx", "Txyz"):
        for op in ("diff", "interp", "min", "max"):
            for ax in ("X", "Y"):
                c.append(("box", op, var, [ax], {}))
        c.append(("box", "interp", var, [["X", "Y"]], {}))
        c.append(("box", "diff", var, [["Y", "X"]], {"padding": {"X": "fill", "Y": "periodic"}, "fill_value": {"X": 1.0}}))
        c.append(("box", "derivative", var, ["X"], {}))
        c.append(("box", "derivative", var, ["Y"], {}))
        c.append(("box", "interp", var, ["X"], {"metric_weighted": ("X", "Y")}))
        c.append(("box", "interp", var, [["X", "Y"]], {"metric_weighted": ("X", "Y")}))
        c.append(("box", "integrate", var, [["X", "Y"]], {}))
        c.append(("box", "average", var, ["X"], {}))
        c.append(("box", "cumint", var, ["Y"], {"padding": "fill"}))
        c.append(("box", "cumsum", var, ["X"], {"to": "left", "padding": "fill"}))
    for op in ("diff", "interp"):
        c.append(("box", op, "T", ["Z"], {}))
        c.append(("box", op, "T", ["Z"], {"to": "outer", "padding": "extend"}))
        c.append(("box", op, "W", ["Z"], {}))
        c.append(("box", op, "U", ["X"], {}))
        c.append(("box", op, "V", ["Y"], {"padding": "fill", "fill_value": -1.0}))
    c += [("box", "derivative", "T", ["Z"], {}), ("box", "derivative", "W", ["Z"], {}), ("box", "integrate", "T", ["Z"], {}),
          ("box", "integrate", "T", [["X", "Y", "Z"]], {}), ("box", "average", "T", ["Z"], {}), ("box", "average", "T", [["X", "Y"]], {}),
          ("box", "cumint", "T", ["Z"], {}), ("box", "cumint", "T", ["Z"], {"to": "outer"}), ("box", "cumsum", "T", ["Z"], {}),
          ("box", "cumsum", "T", ["Z"], {"to": "outer", "reverse": True}), ("box", "cumsum", "T", [["X", "Z"]], {"padding": "fill"}),
          ("box", "integrate", "U", ["X"], {}), ("box", "integrate", "V", [["X", "Y"]], {}), ("box", "average", "U", [["X", "Y"]], {}),
          ("box", "derivative", "U", ["X"], {}), ("box", "derivative", "V", ["Y"], {}), ("box", "derivative", "U", ["Y"], {}),
          ("box", "interp_like", "T", ["U"], {}), ("box", "interp_like", "U", ["T"], {}), ("box", "interp_like", "T", ["W"], {"padding": "extend"}),
          ("box", "get_metric", "T", [("X",)], {}), ("box", "get_metric", "U", [("X",)], {}), ("box", "get_metric", "T", [("X", "Y")], {}),
          ("box", "get_metric", "U", [("X", "Y")], {}), ("box", "get_metric", "T", [("Z",)], {}), ("box", "get_metric", "W", [("Z",)], {}),
          ("box", "get_metric", "T", [("X", "Y", "Z")], {}), ("box", "get_metric", "V", [("Y", "Z")], {}),
          ("box", "diff", "T", ["X"], {"boundary": "fill"}), ("box", "diff", "T", ["X"], {"keep_coords": True})]
    for name in USER_UFUNCS:
        c.append(("box", "apply_as_grid_ufunc:" + name, "T", [], {}))
        c.append(("box", "apply_as_grid_ufunc:" + name, "Tyx", [], {}))
    c += [("box", "integrate", "T", ["Z"], {"skipna": False}), ("box", "average", "T", ["Z"], {"skipna": False}),
          ("box", "integrate", "T", ["X"], {"keep_attrs": True}), ("box", "cumsum", "T", ["X"], {"to": "left", "metric_weighted": "X", "padding": "fill"})]
    for ds in ("faces_x2x", "faces_x2y", "faces_x2y_rev", "cube"):
        for op in ("diff", "interp"):
            for ax in ("X", "Y"):
                c.append((ds, op, "c", [ax], {}))
                c.append((ds, op, "c", [ax], {"padding": "extend"}))
            c.append((ds, op, "u", ["X"], {}))                                      # a bare component: scalar halos
            c.append((ds, op, "vec:X:u", ["X"], {"other_component": "vec:Y:v"}))      # vector-aware halos
            c.append((ds, op, "vec:Y:v", ["Y"], {"other_component": "vec:X:u"}))
            c.append((ds, op, "vec:Y:v", ["X"], {"other_component": "vec:X:u"}))
            c.append((ds, op, "vec:X:u", ["Y"], {"other_component": "vec:Y:v"}))
        for ax, var in (("X", "c"), ("Y", "c"), ("X", "u"), ("Y", "v")):
            for reverse in (False, True):
                c.append((ds, "cumsum", var, [ax], {"reverse": reverse}))
        c.append((ds, "diff", "vec:X:u", ["X"], {}))                                  # vector without its partner
    for op in ("diff", "interp"):
        for var in ("t", "u", "v"):
            for ax in ("X", "Y"):
                c.append(("fold", op, var, [ax], {}))
        c.append(("fold", op, "vec:X:u", ["Y"], {"other_component": "vec:Y:v"}))
        c.append(("fold", op, "vec:Y:v", ["Y"], {"other_component": "vec:X:u"}))
    return c


def describe(res):
    return {"dims": list(res.dims), "name": res.name, "coords": sorted(res.coords), "dtype": str(res.dtype)}


def main():
    xr, refgrid = import_reference()
    rng = np.random.default_rng(20260927)
    dss = datasets(rng)
    arrays, meta = {}, {"datasets": {}, "calls": []}
    built = {}
    for name, (variables, coords, gkw) in dss.items():
        ds = xr.Dataset({k: v for k, v in variables.items()}, coords)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            grid = refgrid.Grid(ds, **gkw)
        built[name] = (ds, grid)
        d = {"variables": {}, "coords": {}, "grid": {k: ({"|".join(kk): vv for kk, vv in v.items()} if k == "metrics" else v) for k, v in gkw.items()}}
        for k, v in variables.items():
            arrays[f"{name}/var/{k}"] = np.asarray(v[1])
            d["variables"][k] = {"dims": list(v[0]), "attrs": v[2] if len(v) > 2 else {}}
        for k, v in coords.items():
            arrays[f"{name}/coord/{k}"] = np.asarray(v[1])
            d["coords"][k] = {"dims": [v[0]] if isinstance(v[0], str) else list(v[0]), "attrs": v[2] if len(v) > 2 else {}}
        meta["datasets"][name] = d
    n_ok = n_err = 0
    for i, (dsname, method, var, args, kw) in enumerate(calls()):
        ds, grid = built[dsname]
        rec = {"id": i, "dataset": dsname, "method": method, "var": var, "args": args, "kwargs": kw}
        call_args = [ds[a] if (method == "interp_like" and isinstance(a, str)) else a for a in args]

        def operand(spec):
            if isinstance(spec, str) and spec.startswith("vec:"):
                _, ax, name = spec.split(":")
                return {ax: ds[name]}
            return ds[spec]

        call_kw = {k: (operand(v) if k == "other_component" else v) for k, v in kw.items()}
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                if method.startswith("apply_as_grid_ufunc:"):
                    func, ukw = USER_UFUNCS[method.split(":")[1]]
                    if var == "Tyx" and method.endswith("second_order_diff") is False and False:
                        pass
                    res = grid.apply_as_grid_ufunc(func, operand(var), **ukw)
                else:
                    res = getattr(grid, method)(operand(var), *call_args, **call_kw)
            rec["warnings"] = sorted({type(w.message).__name__ + ": " + str(w.message)[:60] for w in caught})
            outs = list(res) if isinstance(res, (tuple, list)) else [res]
            rec["result"] = describe(outs[0])
            rec["more_results"] = [describe(o) for o in outs[1:]]
            for k, o in enumerate(outs):
                key = f"call/{i}" if k == 0 else f"call/{i}/out{k}"
                arrays[key] = np.asarray(o.values)
                for cname in o.coords:
                    arrays[f"{key}/coord/{cname}"] = np.asarray(o.coords[cname].values)
            n_ok += 1
        except Exception as exc:  # noqa: BLE001 -- the error IS the expected behaviour
            rec["raises"] = {"type": type(exc).__name__, "message": str(exc)[:160]}
            n_err += 1
        meta["calls"].append(rec)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "grid_reference.npz"), **arrays)
    with open(os.path.join(OUT, "grid_reference.json"), "w") as f:
        json.dump(meta, f, indent=0, default=lambda o: list(o) if isinstance(o, tuple) else str(o))
    print(f"{n_ok} results, {n_err} expected errors -> tests/golden/grid_reference.{{npz,json}}")
    by = {}
    for r in meta["calls"]:
        if "raises" in r:
            by.setdefault(r["raises"]["type"] + ": " + r["raises"]["message"][:70], []).append(r["id"])
    for k, v in by.items():
        print("  raises", k, "x", len(v))


if __name__ == "__main__":
    main()
