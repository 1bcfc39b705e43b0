#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (it needs /root/reference).  It never runs on the GPU
box and nothing it imports ships: only the *data* it writes is committed.

How the reference is executed here
----------------------------------
xarray and dask are not installed (and cannot be), so `import xgcm` fails.  The reference's
pure-numpy pieces however only need the *names* ``xarray.DataArray/Dataset`` and
``dask.array.Array`` to exist for type annotations / isinstance checks.  We register empty
placeholder modules carrying those three names -- no behaviour is stubbed, nothing of
xarray is re-implemented -- and then import the reference's own modules and execute:

  * every ``GridUFunc.ufunc`` body in ``xgcm/gridops.py`` (41 of them) on seeded arrays that
    were padded with ``numpy.pad`` (the routine ``DataArray.pad`` forwards to): float64 inputs
    -> ``gridops_vectors.npz``; bool / int8..int64 / uint8..uint64 inputs with wrap-around,
    2^53 + 1 and 2^62 cases -> ``gridops_vectors_int.npz``;
  * ``_GridUFuncSignature.from_string / equivalent / __str__`` (xgcm/grid_ufunc.py:147-301);
  * ``_select_grid_ufunc`` (xgcm/grid.py:1779-1824);
  * ``iterate_axis_combinations`` (xgcm/metrics.py:4-30);
  * the north-fold index helpers ``_seam_partner_indices``, ``_resolve_pivot`` and
    ``_parse_fold_padding`` (xgcm/padding.py:94-177) -> ``fold_reference.json``;
  * the ``cases`` data table of xgcm/test/test_transform.py:40-686 (inputs + expected outputs of
    the transform tests) -> ``transform_cases.json``.

The xarray-level glue (apply_as_grid_ufunc, pad, Grid.cumsum ...) can NOT be executed; for
it we transcribe the deterministic known-answer tests of the reference's own test-suite
(inputs and expected outputs, each with its file:line) into ``kats.json``.
"""

from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
REF = "/root/reference"


def _import_reference():
    xr = types.ModuleType("xarray")
    xr.DataArray = type("DataArray", (), {})
    xr.Dataset = type("Dataset", (), {})
    sys.modules["xarray"] = xr
    dask = types.ModuleType("dask")
    dask_array = types.ModuleType("dask.array")
    dask_array.Array = type("Array", (), {})
    dask.array = dask_array
    sys.modules["dask"] = dask
    sys.modules["dask.array"] = dask_array
    sys.path.insert(0, REF)
    import xgcm.grid as grid  # noqa
    import xgcm.grid_ufunc as grid_ufunc  # noqa
    import xgcm.gridops as gridops  # noqa
    import xgcm.metrics as metrics  # noqa

    return gridops, grid_ufunc, grid, metrics


def dispatch_table(gridops, grid_ufunc):
    rows = []
    for name, obj in vars(gridops).items():
        if isinstance(obj, grid_ufunc.GridUFunc):
            rows.append(
                {
                    "name": name,
                    "signature": str(obj.signature),
                    "padding_width": {k: list(v) for k, v in (obj.padding_width or {}).items()}
                    if obj.padding_width is not None
                    else None,
                    "padding": obj.padding,
                    "fill_value": obj.fill_value,
                    "dask": obj.dask,
                    "map_overlap": obj.map_overlap,
                    "pad_before_func": obj.pad_before_func,
                }
            )
    return rows


def gridops_vectors(gridops, grid_ufunc):
    """Outputs of every reference ufunc body on seeded, numpy-padded inputs."""
    rng = np.random.default_rng(20260926)
    out = {}
    modes = {"periodic": "wrap", "fill": "constant", "extend": "edge"}
    for name, obj in vars(gridops).items():
        if not isinstance(obj, grid_ufunc.GridUFunc) or name == "diff_left_to_inner":
            continue
        a = rng.standard_normal((3, 4, 11))
        if name.startswith(("min", "max")):
            a[1, 2, 5] = np.nan  # NaN propagation of np.min/np.max (gridops.py:123-175)
            a[0, 0, 0] = np.nan
            a[2, 3, 10] = np.nan
        out[f"{name}|in"] = a
        lo, hi = obj.padding_width["X"]
        if obj.pad_before_func:
            for bc, mode in modes.items():
                kw = {"constant_values": 1.25} if mode == "constant" else {}
                p = np.pad(a, [(0, 0), (0, 0), (lo, hi)], mode, **kw)
                out[f"{name}|{bc}"] = obj.ufunc(p)
        else:  # pad after func (cumsum family, gridops.py:221-278)
            r = obj.ufunc(a)
            for bc, mode in modes.items():
                kw = {"constant_values": 1.25} if mode == "constant" else {}
                out[f"{name}|{bc}"] = np.pad(r, [(0, 0), (0, 0), (lo, hi)], mode, **kw)
    return out


INT_DTYPES = ("bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64")
INT_FILL = 3.7  # numpy.pad casts the constant to the array's dtype: 3 (True for bool)


def int_field(rng, dtype, shape):
    """seeded integer field with values the float64 lanes would get wrong: the dtype's extremes (wrap-around of
    diff / interp sums / cumsum), and for the 64-bit types neighbours of 2^53 and 2^62 (not representable in float64)"""
    dt = np.dtype(dtype)
    if dt == np.bool_:
        return rng.integers(0, 2, shape).astype(dt)
    info = np.iinfo(dt)
    a = rng.integers(info.min, info.max, shape, dtype=dt, endpoint=True)
    flat = a.reshape(-1)
    flat[0], flat[-1] = info.max, info.min
    if dt.itemsize == 8:
        specials = [2**53 + 1, 2**53 + 3, 2**62, 2**62 + 1, 2**53 + 5]
        if dt.kind == "i":
            specials += [-(2**53 + 1), -(2**62) - 7]
        else:
            specials += [2**63 + 11, 2**64 - 2]
        for k, v in enumerate(specials):
            flat[3 + 2 * k] = v
    return a


def gridops_vectors_int(gridops, grid_ufunc):
    """Outputs of every reference ufunc body on seeded INTEGER / bool inputs (numpy keeps them integral through diff /
    min / max / cumsum / pad and wraps; interp leaves through `/ 2.0`).  A body that raises (diff of bool: numpy refuses
    boolean subtract) is recorded as a 0-d string array holding the exception's class name."""
    rng = np.random.default_rng(20260927)
    out = {}
    modes = {"periodic": "wrap", "fill": "constant", "extend": "edge"}
    for name, obj in vars(gridops).items():
        if not isinstance(obj, grid_ufunc.GridUFunc) or name == "diff_left_to_inner":
            continue
        lo, hi = obj.padding_width["X"]
        for dtype in INT_DTYPES:
            a = int_field(rng, dtype, (3, 4, 11))
            out[f"{name}|{dtype}|in"] = a
            for bc, mode in modes.items():
                kw = {"constant_values": INT_FILL} if mode == "constant" else {}
                try:
                    if obj.pad_before_func:
                        r = obj.ufunc(np.pad(a, [(0, 0), (0, 0), (lo, hi)], mode, **kw))
                    else:  # pad after func (cumsum family, gridops.py:221-278): the pad acts on the cumulative values
                        r = np.pad(obj.ufunc(a), [(0, 0), (0, 0), (lo, hi)], mode, **kw)
                except TypeError as e:
                    r = np.array(type(e).__name__)
                out[f"{name}|{dtype}|{bc}"] = r
    return out


def signature_cases(grid_ufunc):
    S = grid_ufunc._GridUFuncSignature
    strings = [
        "()->()",
        "(X:center)->()",
        "()->(X:left)",
        "(X:center)->(X:left)",
        "(X:left)->(Y:center)",
        "(X:left),(X:right)->(Y:center)",
        "(X:center)->(Y:inner),(Y:outer)",
        "(X:center,Y:center)->(Z:center)",
        "(X:center, Y:center) -> (X:left,Y:center),(X:center,Y:left)",
        "(lon:center,lat:center)->(lon:left,lat:center),(lon:center,lat:left)",
        # invalid (test/test_grid_ufunc.py:84-101)
        "(x:left)(y:left)->()",
        "(x:left),(y:left)->",
        "((x:left))->(x:left)",
        "(x:left)->(x:left),",
        "(i)->(i)",
        "(X:centre)->()",
    ]
    parse = []
    for s in strings:
        try:
            sig = S.from_string(s)
            parse.append(
                {
                    "string": s,
                    "ok": True,
                    "in_ax_names": [list(t) for t in sig.in_ax_names],
                    "in_ax_positions": [list(t) for t in sig.in_ax_positions],
                    "out_ax_names": [list(t) for t in sig.out_ax_names],
                    "out_ax_positions": [list(t) for t in sig.out_ax_positions],
                    "str": str(sig),
                }
            )
        except ValueError as e:
            parse.append({"string": s, "ok": False, "error": str(e)})
    pairs = [
        ("(X:center)->(X:left)", "(Z:center)->(Z:left)"),
        ("(X:center)->(X:left)", "(X:center)->(X:right)"),
        ("(X:center)->(X:left)", "(X:center)->(Y:left)"),
        # NB: multi-axis pairs are deliberately absent: the reference zips two *sets* of dummy
        # names (grid_ufunc.py:243-259), so its answer for >1 dummy axis depends on
        # PYTHONHASHSEED.  Single-axis signatures (all the built-in dispatch uses) are stable.
        ("(X:center),(X:left)->(X:left)", "(Y:center),(Y:left)->(Y:left)"),
        ("(X:center)->()", "(Y:center)->()"),
        ("(X:center)->(X:center)", "(X:center),(X:center)->(X:center)"),
    ]
    equiv = []
    for a, b in pairs:
        equiv.append({"a": a, "b": b, "equivalent": bool(S.from_string(a).equivalent(S.from_string(b)))})
    return {"parse": parse, "equivalent": equiv}


def select_cases(gridops, grid_ufunc, grid):
    S = grid_ufunc._GridUFuncSignature
    name_of = {id(v): k for k, v in vars(gridops).items() if isinstance(v, grid_ufunc.GridUFunc)}
    positions = ["center", "left", "right", "inner", "outer"]
    rows = []
    for funcname in ["diff", "interp", "min", "max", "cumsum", "nosuchfunc"]:
        for f in positions:
            for t in positions:
                sig = S.from_string(f"(Q:{f})->(Q:{t})")
                try:
                    uf, rest = grid._select_grid_ufunc(funcname, sig, module=gridops, padding="fill")
                    rows.append(
                        {"funcname": funcname, "from": f, "to": t, "selected": name_of[id(uf)], "kwargs": rest}
                    )
                except (NotImplementedError, ValueError) as e:
                    rows.append(
                        {"funcname": funcname, "from": f, "to": t, "error": type(e).__name__, "message": str(e)}
                    )
    return rows


def metrics_cases(metrics):
    rows = []
    for items in (["X"], ["X", "Y"], ["X", "Y", "Z"]):
        combos = [sorted(sorted(fs) for fs in combo) for combo in metrics.iterate_axis_combinations(items)]
        # set iteration order inside the reference depends on PYTHONHASHSEED; the first yield
        # (the full set) is order-stable, the rest is recorded as an order-free collection.
        rows.append({"items": items, "first": combos[0], "rest_sorted": sorted(combos[1:])})
    return rows


def kats():
    """Deterministic known-answer tests transcribed (inputs + expected outputs) from the
    reference's test-suite and docs.  `grid` describes the axis topology; `call` the public
    Grid method invocation; values are float64."""
    k = []
    five = lambda ax, n: {  # test/test_grid_ufunc.py:216-268 (create_1d_test_grid)
        "center": [f"{ax}_c", n],
        "left": [f"{ax}_g", n],
        "right": [f"{ax}_r", n],
        "inner": [f"{ax}_i", n - 1],
        "outer": [f"{ax}_o", n + 1],
    }
    # 1. test/test_grid_ufunc.py:1312-1338
    k.append(
        {
            "name": "interp_center_to_outer_extend",
            "source": "xgcm/test/test_grid_ufunc.py:1312-1338",
            "grid": {"axes": {"Z": {"center": ["Z", 10], "outer": ["Zp1", 11]}}, "padding": None},
            "call": {"method": "interp", "axis": "Z", "kwargs": {"padding": "extend", "to": "outer"}},
            "in_dims": ["Z"],
            "in": np.linspace(1, 10, num=10).tolist(),
            "out_dims": ["Zp1"],
            "out": np.concatenate(([1.0], np.linspace(1.5, 9.5, num=9), [10.0])).tolist(),
            "exact": True,
        }
    )
    # 2. test/test_grid_ufunc.py:1214-1273 (user ufunc interp c->l with fill 0 / 1 / 10)
    arr = np.arange(9).astype(float)
    for fv in (0, 1, 10):
        p = np.concatenate([[float(fv)], arr])
        k.append(
            {
                "name": f"interp_center_to_left_fill_{fv}",
                "source": "xgcm/test/test_grid_ufunc.py:1214-1273",
                "grid": {"axes": {"lat": five("lat", 9)}, "padding": "periodic"},
                "call": {"method": "interp", "axis": "lat", "kwargs": {"padding": "fill", "fill_value": fv, "to": "left"}},
                "in_dims": ["lat_c"],
                "in": arr.tolist(),
                "out_dims": ["lat_g"],
                "out": (0.5 * (p[:-1] + p[1:])).tolist(),
                "exact": True,
            }
        )
    # 3. test/test_grid.py:528-552
    zl = np.arange(1.0, 15.0)
    k.append(
        {
            "name": "cumsum_center_to_outer_fill0",
            "source": "xgcm/test/test_grid.py:528-552",
            "grid": {"axes": {"Z": {"center": ["zl", 14], "outer": ["zi", 15]}}, "padding": "fill"},
            "call": {"method": "cumsum", "axis": "Z", "kwargs": {"padding": "fill", "fill_value": 0.0}},
            "in_dims": ["zl"],
            "in": zl.tolist(),
            "out_dims": ["zi"],
            "out": np.hstack([0.0, np.cumsum(zl)]).tolist(),
            "exact": True,
        }
    )
    k.append(
        {
            "name": "cumsum_no_boundary_raises",
            "source": "xgcm/test/test_grid.py:541-545",
            "grid": {"axes": {"Z": {"center": ["zl", 14], "outer": ["zi", 15]}}, "padding": None},
            "call": {"method": "cumsum", "axis": "Z", "kwargs": {}},
            "in_dims": ["zl"],
            "in": zl.tolist(),
            "raises": "ValueError",
            "match": "No boundary condition was specified",
        }
    )
    # 4. test/test_grid_ufunc.py:941-967 (pad-after-func cumsum center->left, fill 0)
    sq = np.arange(1, 10).astype(float) ** 2
    c = np.roll(np.cumsum(sq), 1)
    c[0] = 0
    k.append(
        {
            "name": "cumsum_center_to_left_pad_after_fill0",
            "source": "xgcm/test/test_grid_ufunc.py:941-967",
            "grid": {"axes": {"depth": five("depth", 9)}, "padding": "periodic"},
            "call": {"method": "cumsum", "axis": "depth", "kwargs": {"padding": "fill", "fill_value": 0, "to": "left"}},
            "in_dims": ["depth_c"],
            "in": sq.tolist(),
            "out_dims": ["depth_g"],
            "out": c.tolist(),
            "exact": True,
        }
    )
    # 5. test/test_metrics_ops.py:135-179: derivative == diff/dx bitwise on both axes, periodic
    a44 = np.array([[1.0, 2.0, 4.0, 3.0], [4.0, 7.0, 1.0, 2.0], [3.0, 1.0, 0.0, 9.0], [8.0, 5.0, 2.0, 1.0]])
    g44 = {
        "axes": {"X": {"center": ["XC", 4], "left": ["XG", 4]}, "Y": {"center": ["YC", 4], "left": ["YG", 4]}},
        "padding": "periodic",
        "metrics": {
            "X": {"dXC": [["XC"], [10.0] * 4], "dXG": [["XG"], [10.0] * 4]},
            "Y": {"dYC": [["YC"], [10.0] * 4], "dYG": [["YG"], [10.0] * 4]},
        },
    }
    k.append(
        {
            "name": "derivative_uniform_X",
            "source": "xgcm/test/test_metrics_ops.py:135-179",
            "grid": g44,
            "call": {"method": "derivative", "axis": "X", "kwargs": {}},
            "in_dims": ["XC", "YC"],
            "in": a44.tolist(),
            "out_dims": ["XG", "YC"],
            "out": ((a44 - np.roll(a44, 1, axis=0)) / 10.0).tolist(),
            "exact": True,
        }
    )
    k.append(
        {
            "name": "derivative_uniform_Y",
            "source": "xgcm/test/test_metrics_ops.py:135-179",
            "grid": g44,
            "call": {"method": "derivative", "axis": "Y", "kwargs": {}},
            "in_dims": ["XC", "YC"],
            "in": a44.tolist(),
            "out_dims": ["XC", "YG"],
            "out": ((a44 - np.roll(a44, 1, axis=1)) / 10.0).tolist(),
            "exact": True,
        }
    )
    # 6. test/test_grid_ufunc.py:1278-1310: diff center->right periodic == roll(-1) - a
    dc = np.arange(1, 10).astype(float)
    s = np.sin(dc * 2 * np.pi / 9)
    k.append(
        {
            "name": "diff_center_to_right_periodic_1d",
            "source": "xgcm/test/test_grid_ufunc.py:1278-1291",
            "grid": {"axes": {"depth": five("depth", 9)}, "padding": "periodic"},
            "call": {"method": "diff", "axis": "depth", "kwargs": {"to": "right"}},
            "in_dims": ["depth_c"],
            "in": s.tolist(),
            "out_dims": ["depth_r"],
            "out": (np.roll(s, -1) - s).tolist(),
            "exact": True,
        }
    )
    yc = np.arange(1, 12).astype(float)
    a2 = dc[:, None] ** 2 + yc[None, :] ** 2
    k.append(
        {
            "name": "diff_center_to_right_periodic_2d_first_axis",
            "source": "xgcm/test/test_grid_ufunc.py:1293-1310",
            "grid": {"axes": {"depth": five("depth", 9), "y": five("y", 11)}, "padding": "periodic"},
            "call": {"method": "diff", "axis": "depth", "kwargs": {"to": "right"}},
            "in_dims": ["depth_c", "y_c"],
            "in": a2.tolist(),
            "out_dims": ["depth_r", "y_c"],
            "out": (np.roll(a2, -1, axis=0) - a2).tolist(),
            "exact": True,
        }
    )
    # 7. test/test_grid_ufunc.py:690-741: diff center->left periodic == a - roll(a, 1)
    k.append(
        {
            "name": "diff_center_to_left_periodic_1d",
            "source": "xgcm/test/test_grid_ufunc.py:690-741",
            "grid": {"axes": {"depth": five("depth", 9)}, "padding": "periodic"},
            "call": {"method": "diff", "axis": "depth", "kwargs": {"to": "left"}},
            "in_dims": ["depth_c"],
            "in": s.tolist(),
            "out_dims": ["depth_g"],
            "out": (s - np.roll(s, 1)).tolist(),
            "exact": True,
        }
    )
    # 8. test/test_grid_ufunc.py:662-688: user ufunc, width (2,0), periodic
    k.append(
        {
            "name": "user_ufunc_second_order_diff_width2_periodic",
            "source": "xgcm/test/test_grid_ufunc.py:662-688",
            "grid": {"axes": {"depth": five("depth", 9)}, "padding": "periodic"},
            "call": {
                "method": "apply_as_grid_ufunc",
                "ufunc": "second_order_diff",
                "axis": [["depth"]],
                "kwargs": {"signature": "(X:center)->(X:center)", "padding_width": {"X": [2, 0]}},
            },
            "in_dims": ["depth_c"],
            "in": s.tolist(),
            "out_dims": ["depth_c"],
            "out": (0.5 * (s - np.roll(s, 2))).tolist(),
            "exact": True,
        }
    )
    # 9. docs/boundary_conditions.md:56-113: last-point diff of g (left -> center) under 4 bcs
    xg = np.arange(0.5, 9)
    g = np.sqrt(xg + 0.5) + np.sin((xg - 0.5) * 2 * np.pi / 8)
    for nm, kw, last in (
        ("extend", {"padding": "extend"}, 0.0),
        ("fill0", {"padding": "fill", "fill_value": 0}, -3.0),
        ("fill5", {"padding": "fill", "fill_value": 5}, 2.0),
        ("periodic", {"padding": "periodic"}, -2.0),
    ):
        k.append(
            {
                "name": f"docs_boundary_last_point_{nm}",
                "source": "docs/boundary_conditions.md:56-113",
                "grid": {"axes": {"X": {"center": ["x_c", 9], "left": ["x_g", 9]}}, "padding": "fill"},
                "call": {"method": "diff", "axis": "X", "kwargs": kw},
                "in_dims": ["x_g"],
                "in": g.tolist(),
                "out_dims": ["x_c"],
                "out_last": last,  # the docs state only the last point, to print precision
                "atol": 1e-12,
            }
        )
    # 10. test/test_grid.py:196-285: cumsum extend/fill semantics, forward and reverse, on a
    #     deterministic stand-in for the (unseeded) nonperiodic_1d fixture: extend replicates the
    #     first (forward) / last (reverse) CUMULATIVE value.
    x = np.array([0.5, -1.25, 2.0, 4.5, -0.75, 1.0, 3.25, -2.5, 0.125])
    cf = np.cumsum(x)
    cr = np.cumsum(x[::-1])[::-1]
    for padding in ("extend", "fill"):
        fv_f = 0.0 if padding == "fill" else cf[0]
        fv_r = 0.0 if padding == "fill" else cr[-1]
        cases = {
            ("center", "left", False): np.hstack([fv_f, cf[:-1]]),
            ("center", "right", False): cf,
            ("center", "outer", False): np.hstack([fv_f, cf]),
            ("center", "inner", False): cf[:-1],
            ("center", "left", True): cr,
            ("center", "right", True): np.hstack([cr[1:], fv_r]),
            ("center", "outer", True): np.hstack([cr, fv_r]),
            ("center", "inner", True): cr[1:],
        }
        for (f, t, rev), exp in cases.items():
            k.append(
                {
                    "name": f"cumsum_{f}_to_{t}_{padding}_{'reverse' if rev else 'forward'}",
                    "source": "xgcm/test/test_grid.py:196-285",
                    "grid": {"axes": {"X": five("X", 9)}, "padding": "periodic"},
                    "call": {"method": "cumsum", "axis": "X", "kwargs": {"to": t, "padding": padding, "reverse": rev}},
                    "in_dims": ["X_c"],
                    "in": x.tolist(),
                    "out_dims": [five("X", 9)[t][0]],
                    "out": exp.tolist(),
                    "exact": True,
                }
            )
    return k


def fold_cases():
    """Outputs of the reference's own north-fold helpers (pure numpy / dict logic)."""
    import xgcm.padding as P

    seam = []
    for position in ("center", "left", "right", "outer", "inner"):
        for pivot_seam in ("edge", "center"):
            for length in (3, 4, 5, 8, 9, 12):
                seam.append({"position": position, "pivot_seam": pivot_seam, "length": length,
                             "indices": P._seam_partner_indices(position, pivot_seam, length).tolist()})
    pivots = []
    for spec in ("center", "T", "t", "corner", "F", "U", "u", "V", {"X": "right", "Y": "center"},
                 {"X": "left", "Y": "left"}, {"X": "right", "Y": "right"}, {"Y": "outer"}, {"X": "inner"},
                 {"X": "center"}):
        pivots.append({"pivot": spec, "fold_axis": "Y", "seam_axis": "X",
                       "roles": P._resolve_pivot(spec, "Y", "X")})
    bad_pivot = {"pivot": {"Z": "left"}, "fold_axis": "Y", "seam_axis": "X"}
    try:
        P._resolve_pivot(bad_pivot["pivot"], "Y", "X")
    except Exception as exc:  # noqa: BLE001
        bad_pivot["raises"] = type(exc).__name__
        bad_pivot["message"] = str(exc)
    pivots.append(bad_pivot)
    parses = []
    for spec in ({"fold": "corner"}, {"fold": "U", "south": "extend"}, {"fold": {"X": "left"}, "south": "periodic"},
                 {"fold": "banana"}, {"fold": "corner", "north": 1}, {"fold": {}}, {"fold": {"X": "centre"}},
                 {"fold": 3}, {"fold": "corner", "south": "wrap"}, {"south": "fill"}):
        row = {"spec": spec}
        try:
            row["parsed"] = P._parse_fold_padding(spec)
        except Exception as exc:  # noqa: BLE001
            row["raises"] = type(exc).__name__
            row["message"] = str(exc)
        parses.append(row)
    return {"seam_partner_indices": seam, "resolve_pivot": pivots, "parse_fold_padding": parses}


def transform_cases():
    """The `cases` table of the reference's transform tests (xgcm/test/test_transform.py:40-686):
    inputs and expected outputs, evaluated here (its expected values call numpy.interp) and written
    out as plain data.  Only that dictionary literal is executed; the module itself needs xarray."""
    path = os.path.join(REF, "xgcm", "test", "test_transform.py")
    with open(path) as f:
        lines = f.read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith("cases = {"))
    stop = next(i for i, ln in enumerate(lines) if ln.startswith("def construct_test_source_data"))
    ns = {"np": np}
    exec("\n".join(lines[start:stop]), ns)  # noqa: S102  (reference test DATA, build container only)

    def plain(v):
        if isinstance(v, np.ndarray):
            return plain(v.tolist())
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        return v

    return {name: plain(case) for name, case in ns["cases"].items()}


def config1(gridops):
    """BASELINE.json configs[0]: Grid.diff along X on a 128 x 64 periodic C-grid (YC=64, XC=128), f64,
    center->left: the reference's own ufunc body on the numpy.pad(wrap)-ed synthetic field (seed 1)."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle.refimpl import synthetic_field

    a = synthetic_field((64, 128), 1)
    p = np.pad(a, [(0, 0), (1, 0)], "wrap")
    return {"in": a, "diff_X_center_to_left_periodic": gridops.diff_center_to_left.ufunc(p),
            "interp_X_center_to_left_periodic": gridops.interp_center_to_left.ufunc(p)}


def main():
    os.makedirs(OUT, exist_ok=True)
    gridops, grid_ufunc, grid, metrics = _import_reference()
    np.savez_compressed(os.path.join(OUT, "config1.npz"), **config1(gridops))
    with open(os.path.join(OUT, "gridops_table.json"), "w") as f:
        json.dump(dispatch_table(gridops, grid_ufunc), f, indent=1)
    np.savez_compressed(os.path.join(OUT, "gridops_vectors.npz"), **gridops_vectors(gridops, grid_ufunc))
    np.savez_compressed(os.path.join(OUT, "gridops_vectors_int.npz"), **gridops_vectors_int(gridops, grid_ufunc))
    with open(os.path.join(OUT, "signatures.json"), "w") as f:
        json.dump(signature_cases(grid_ufunc), f, indent=1)
    with open(os.path.join(OUT, "select.json"), "w") as f:
        json.dump(select_cases(gridops, grid_ufunc, grid), f, indent=1)
    with open(os.path.join(OUT, "metrics_combos.json"), "w") as f:
        json.dump(metrics_cases(metrics), f, indent=1)
    with open(os.path.join(OUT, "kats.json"), "w") as f:
        json.dump(kats(), f, indent=1)
    with open(os.path.join(OUT, "fold_reference.json"), "w") as f:
        json.dump(fold_cases(), f, indent=1)
    with open(os.path.join(OUT, "transform_cases.json"), "w") as f:
        json.dump(transform_cases(), f, indent=1)
    print("wrote fixtures to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
