#!/usr/bin/env python3
"""Differential fuzzing of `xgcm_amd.Grid` against the REFERENCE's own `Grid`, live, call by call.

TEST INFRASTRUCTURE -- build container only (imports /root/reference at run time), never shipped, never on the GPU box.

Both stacks run in ONE process over the same stand-in `xarray` (`oracle/xr_min.py` loaded under that name): the reference's
`xgcm/grid.py`, `axis.py`, `grid_ufunc.py`, `padding.py`, `gridops.py`, `metrics.py` imported unmodified as package `xgcm`,
and `xgcm_amd` with its device served by a CPU double.  A seeded generator draws a grid (1-3 axes, 2-3 positions per axis,
random lengths, boundary conditions and fill values as strings / per-axis dicts / unset, metrics at some positions), a
handful of variables (random positions, dim orders, an extra record dim, NaNs, float32 / integer fields now and then) and
random calls -- diff / interp / min / max / cumsum / derivative / integrate / average / cumint / interp_like / get_metric
with random `axis`, `to`, `padding`, `fill_value`, `metric_weighted`, `reverse`, `skipna`, some of them invalid -- and every
call is made on both grids.  They must agree on: raising or not; the exception type (the message too, reported separately);
result dims, name, dtype, coordinate names and values; the values bit for bit, except contiguous-axis scans and reductions
of the product, which are compared to 1e-12.

    python oracle/fuzz_against_reference.py --cases 200 --seed 1 [--backend oracle-double|host-abi] [-v]

`tests/test_reference_suite_live.py::test_differential_fuzz_against_the_reference` runs a fixed budget of it on every CPU run.
What it pins is the reference's LOGIC under the stand-in's container semantics ("pinned modulo the stand-in").
"""
import argparse
import importlib.util
import itertools
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("XGCM_REFERENCE", "/root/reference")

POSITIONS = ("center", "left", "right", "inner", "outer")
LENGTH = {"center": 0, "left": 0, "right": 0, "inner": -1, "outer": +1}
MODES = ("periodic", "fill", "extend")


class _MP:
    def setattr(self, target, name=None, value=None, raising=True):
        setattr(target, name, value)

    def setitem(self, mapping, key, value):
        mapping[key] = value


def load_both(backend="oracle-double"):
    """(xarray stand-in, reference Grid class, xgcm_amd Grid class)"""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    real = None
    if os.environ.get("XGCM_USE_STANDIN") != "1":
        try:  # the real package where it is installed (with the real dask and numba it asks for); this image has none
            import xarray as real  # noqa: F811
        except ImportError:
            real = None
    if real is not None and not getattr(real, "_is_xr_min", False):
        if REF not in sys.path:
            sys.path.append(REF)
        import xgcm.grid as refgrid
        import xgcm_amd

        if backend == "oracle-double":
            from oracle import fake_device

            fake_device.install(_MP())
        elif backend == "host-abi":
            import host_abi_device

            host_abi_device.install(_MP())
        return real, refgrid.Grid, xgcm_amd.Grid
    if "xarray" not in sys.modules or not getattr(sys.modules["xarray"], "_is_xr_min", False):
        spec = importlib.util.spec_from_file_location("xarray", os.path.join(HERE, "xr_min.py"))
        xr = importlib.util.module_from_spec(spec)
        sys.modules["xarray"] = xr
        spec.loader.exec_module(xr)
        xr._is_xr_min = True
        dask = types.ModuleType("dask")
        dask_array = types.ModuleType("dask.array")
        dask_array.Array = type("Array", (), {})
        dask.array = dask_array
        sys.modules.setdefault("dask", dask)
        sys.modules.setdefault("dask.array", dask_array)
    xr = sys.modules["xarray"]
    if "numba" not in sys.modules:
        # `xgcm/transform.py` compiles its two column kernels with numba's `guvectorize`; numba is absent, their bodies are
        # plain Python: the stand-in of oracle/make_golden_transform.py runs them column by column over the loop dims
        from oracle.make_golden_transform import _Type, guvectorize

        nb = types.ModuleType("numba")
        nb.boolean = nb.float32 = nb.float64 = _Type()
        nb.guvectorize = guvectorize
        sys.modules["numba"] = nb
    # (the reference rechunks an interpolated `target_data` to ONE chunk along the axis, xgcm/transform.py:503-505: on a
    # numpy-backed array that is the identity)
    xr.DataArray.chunk = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.append(REF)  # (after the repo: nothing of the repo is called `xgcm`)
    import xgcm.grid as refgrid  # the reference, unmodified

    import xgcm_amd

    if backend == "host-abi":
        import host_abi_device

        host_abi_device.install(_MP())
    elif backend == "oracle-double":
        from oracle import fake_device

        fake_device.install(_MP())
    elif backend != "hip":
        raise ValueError(backend)
    return xr, refgrid.Grid, xgcm_amd.Grid


# ---- the generator -------------------------------------------------------------------------------------------------------
def draw_grid(rng):
    axes = sorted(rng.choice(["X", "Y", "Z"], size=rng.integers(1, 4), replace=False).tolist())
    coords, positions, sizes = {}, {}, {}
    # one axis with rows long enough for the vector / workgroup kernels -- on grids of one or two axes only, and shorter on two:
    # two operator results on different positions of EVERY axis are an outer product over all of them (sizes multiply)
    long_axis = _pick(rng, axes) if (rng.random() < 0.15 and len(axes) <= 2) else None
    for ax in axes:
        # (516 / 1030 only on one-axis grids: an operator result next to a field on another position of the long axis is an
        # outer product -- 1030 x 1031 x the other dims must stay megabytes, on the GPU box too)
        n = int(rng.integers(3, 8)) if ax != long_axis else int(_pick(rng, [33, 64, 130] + ([257, 516, 1030] if len(axes) == 1 else [])))
        extra = rng.choice(POSITIONS[1:], size=rng.integers(1, 3), replace=False).tolist()
        if "inner" in extra and "outer" in extra and rng.random() < 0.5:
            extra.remove("inner")
        pos = {"center": f"{ax.lower()}_c"}
        for p in extra:
            pos[p] = f"{ax.lower()}_{p[0]}"
        if rng.random() < 0.3:  # positions listed in another order: the default shift follows the table, not the listing
            pos = dict(reversed(list(pos.items())))
        positions[ax] = pos
        for p, dim in pos.items():
            m = n + LENGTH[p]
            sizes[dim] = m
            coords[dim] = (dim, np.arange(m) + {"center": 0.5, "left": 0.0, "right": 1.0, "inner": 1.0, "outer": 0.0}[p])
    sizes["t"] = 2
    coords["t"] = ("t", np.array([0.0, 10.0])) if rng.random() < 0.7 else ("t", np.array(["2001-01-01", "2001-02-01"], dtype="datetime64[ns]"))
    coords["label"] = ("t", np.array([3, 4]))
    if rng.random() < 0.4:  # a scalar coordinate and a 2-D one (on two centre dims, or centre x record): they ride along where their dims survive
        coords["ref_level"] = ((), np.float64(5.0))
        a0 = axes[0]
        two = (positions[a0]["center"], positions[axes[1]]["center"]) if len(axes) > 1 else ("t", positions[a0]["center"])
        coords["depth2d"] = (two, np.arange(sizes[two[0]] * sizes[two[1]], dtype=np.float32).reshape(sizes[two[0]], sizes[two[1]]), {"units": "m"})
    kw = {"coords": positions, "autoparse_metadata": False}
    style = rng.integers(0, 4) if rng.random() < 0.5 else 0
    if style == 0:
        kw["padding"] = str(rng.choice(MODES))
    elif style == 1:
        kw["padding"] = {ax: str(rng.choice(MODES)) for ax in axes if rng.random() < 0.8}
    elif style == 2:
        kw["padding"] = {ax: str(rng.choice(MODES)) for ax in axes}
        kw["fill_value"] = {ax: float(rng.integers(-3, 4)) for ax in axes if rng.random() < 0.6}
    if style == 3 and rng.random() < 0.5:
        kw["fill_value"] = float(rng.integers(1, 5))
    r = rng.random()
    if r < 0.12:  # default shifts of the user's (valid ones; now and then a position the axis lacks)
        ax = _pick(rng, axes)
        others = [p for p in positions[ax] if p != "center"]
        shifts = {"center": _pick(rng, others) if rng.random() < 0.85 else str(rng.choice(POSITIONS))}
        if rng.random() < 0.5:
            shifts[_pick(rng, others)] = "center"
        kw["default_shifts"] = {ax: shifts}
    elif r < 0.15:
        kw["padding"] = "cyclic"  # not a mode
    elif r < 0.17:
        kw["padding"] = {"Q": "fill"}  # not an axis
    elif r < 0.19:
        kw["periodic"] = True  # removed argument
    elif r < 0.21:
        kw["boundary"] = "fill"  # renamed argument
    elif r < 0.23:
        ax = _pick(rng, axes)  # (the Grid is told of a dim the dataset lacks; the variables keep the real ones)
        kw["coords"] = dict(positions, **{ax: dict(positions[ax], **{_pick(rng, list(positions[ax])): "no_such_dim"})})
    elif r < 0.25:
        ax = _pick(rng, axes)  # a coordinate of the wrong length for its position
        p = _pick(rng, [q for q in positions[ax] if q != "center"])
        dim = positions[ax][p]
        sizes[dim] += 2
        coords[dim] = (dim, np.arange(sizes[dim]) * 1.0)
    if r >= 0.25 and rng.random() < 0.08:
        # the grid from the dataset's COMODO attributes instead of `coords=` (autoparse_metadata, the reference's default):
        # every dim names its axis and its shift; what position a length and a shift make is for the two parsers to say
        shift = {"left": -0.5, "outer": -0.5, "right": 0.5, "inner": 0.5}
        for ax in axes:
            for p, dim in positions[ax].items():
                attrs = {"axis": ax} if p == "center" else {"axis": ax, "c_grid_axis_shift": shift[p] if rng.random() < 0.9 else 1.0}
                coords[dim] = (coords[dim][0], coords[dim][1], attrs)
        del kw["coords"], kw["autoparse_metadata"]
        kw["_autoparsed"] = True
    if long_axis is not None:
        # every field and metric takes ONE position on the long axis: two positions of it in one expression are an outer
        # product of 1000 x 1000 x ... cells -- gigabytes, on either side
        kw["_long"] = (long_axis, _pick(rng, list(positions[long_axis])))
    return axes, positions, sizes, coords, kw


def draw_variables(rng, axes, positions, sizes, long=None):
    variables, where = {}, {}
    for i in range(int(rng.integers(3, 6))):
        used = [ax for ax in axes if rng.random() < 0.8] or [axes[0]]
        pos = {ax: str(rng.choice(list(positions[ax]))) for ax in used}
        if long is not None and long[0] in pos:
            pos[long[0]] = long[1]
        dims = [positions[ax][pos[ax]] for ax in used]
        if rng.random() < 0.5:
            dims = ["t"] + dims
        if rng.random() < 0.3:
            dims = list(rng.permutation(dims))
        shape = tuple(sizes[d] for d in dims)
        a = rng.standard_normal(shape)
        kind = rng.random()
        if kind < 0.25:
            a.reshape(-1)[rng.integers(0, a.size, size=max(1, a.size // 7))] = np.nan
        elif kind < 0.35:
            a = a.astype(np.float32)
        elif kind < 0.42:
            a = rng.integers(-9, 9, size=shape).astype(np.int64)
        elif kind < 0.50:  # what files hold: other byte orders, narrow integers, bool (numpy computes them as they are)
            a = _pick(rng, [lambda: a.astype(">f8"), lambda: a.astype(">f4"), lambda: rng.integers(-9, 9, size=shape).astype(np.int32),
                            lambda: rng.integers(0, 200, size=shape).astype(np.uint8), lambda: rng.integers(-9, 9, size=shape).astype(">i4"),
                            lambda: rng.integers(0, 9, size=shape).astype(np.uint32), lambda: rng.random(shape) < 0.5,
                            lambda: rng.integers(-90, 90, size=shape).astype(np.int16), lambda: a.astype(np.float16)])()
        layout = rng.random()
        if layout < 0.08:
            a = np.asfortranarray(a)  # column-major memory (what a transposed model output is)
        elif layout < 0.14:
            a = a[..., ::-1]  # a reversed view: negative stride
        elif layout < 0.20:
            wide = np.repeat(a, 2, axis=-1)
            a = wide[..., ::2]  # every second element of a wider buffer: a strided view
        name = f"v{i}"
        variables[name] = (tuple(dims), a) if rng.random() < 0.7 else (tuple(dims), a, {"units": "K", "long_name": name})
        where[name] = pos
    return variables, where


def draw_metrics(rng, axes, positions, sizes, variables, long=None):
    metrics = {}
    for r in (1, 2, 3):
        for combo in itertools.combinations(axes, r):
            if rng.random() > (0.85 if r == 1 else 0.35):
                continue
            names = []
            for k in range(int(rng.integers(1, 4))):
                dims = [positions[ax][str(rng.choice(list(positions[ax]))) if (long is None or ax != long[0]) else long[1]] for ax in combo]
                others = [ax for ax in axes if ax not in combo and rng.random() < 0.3]
                dims += [positions[ax]["center" if (long is None or ax != long[0]) else long[1]] for ax in others]
                name = "m_" + "".join(combo).lower() + f"_{k}"
                variables[name] = (tuple(dims), rng.random(tuple(sizes[d] for d in dims)) + 0.5)
                names.append(name)
            metrics[combo] = names
    return metrics


def _pick(rng, seq):
    return seq[int(rng.integers(0, len(seq)))]


def _axis_arg(rng, axes, present, allow_absent=0.05):
    pool = present if (present and rng.random() > allow_absent) else axes
    k = int(rng.integers(1, min(len(pool), 3) + 1))
    picked = rng.choice(pool, size=k, replace=False).tolist()
    if k == 1 and rng.random() < 0.6:
        return picked[0]
    return picked


def _to_arg(rng, positions, where_var, ax):
    """a target position: mostly one the reference has a ufunc for from the field's position on that axis, sometimes any"""
    here = where_var.get(ax)
    fits = [p for p in positions[ax] if p != here and (here == "center" or p == "center")]
    if fits and rng.random() < 0.85:
        return _pick(rng, fits)
    return str(rng.choice(POSITIONS))


def _padding_arg(rng, axes, kw):
    r = rng.random()
    if r < 0.45:
        return
    if r < 0.75:
        kw["padding"] = str(rng.choice(MODES))
    else:
        kw["padding"] = {ax: str(rng.choice(MODES)) for ax in axes if rng.random() < 0.7}
    r = rng.random()
    if r < 0.25:
        kw["fill_value"] = float(rng.integers(-2, 3)) + 0.5
    elif r < 0.4:
        kw["fill_value"] = {ax: float(rng.integers(-2, 3)) for ax in axes if rng.random() < 0.7}


def _ufunc_same(a):
    return a * 2.0


def _ufunc_centered(a):  # width (1, 1), output as long as the input
    return a[..., 2:] - 2.0 * a[..., 1:-1] + a[..., :-2]


def _ufunc_forward(a):  # width (0, 1)
    return a[..., 1:] - a[..., :-1]


def _ufunc_backward(a):  # width (1, 0)
    return 0.5 * (a[..., 1:] + a[..., :-1])


def _ufunc_wide(a):  # width (2, 1)
    return a[..., 3:] + a[..., :-3]


def _ufunc_two_outputs(a):
    return a[..., 1:] - a[..., :-1], a[..., 1:] + a[..., :-1]


def _ufunc_two_axes(a):  # (Y, X) with width (1, 0) on both
    return a[..., 1:, 1:] - a[..., :-1, :-1]


def _ufunc_reduce(a):
    return a.sum(axis=-1)


USER_UFUNCS = {
    "same": (_ufunc_same, "({A}:{p})->({A}:{p})", None),
    "centered": (_ufunc_centered, "({A}:{p})->({A}:{p})", {"{A}": (1, 1)}),
    "forward": (_ufunc_forward, "({A}:{p})->({A}:{q})", {"{A}": (0, 1)}),
    "backward": (_ufunc_backward, "({A}:{p})->({A}:{q})", {"{A}": (1, 0)}),
    "wide": (_ufunc_wide, "({A}:{p})->({A}:{p})", {"{A}": (2, 1)}),
    "two_outputs": (_ufunc_two_outputs, "({A}:{p})->({A}:{q}),({A}:{q})", {"{A}": (1, 0)}),
    "reduce": (_ufunc_reduce, "({A}:{p})->()", None),
    "untrimmed": (_ufunc_same, "({A}:{p})->({A}:{p})", {"{A}": (1, 0)}),  # forgets to trim its padding: both must refuse
}


def draw_user_ufunc(rng, axes, positions, where_var, present):
    name = _pick(rng, list(USER_UFUNCS))
    func, sig, widths = USER_UFUNCS[name]
    ax = _pick(rng, present) if (present and rng.random() < 0.95) else _pick(rng, axes)
    here = where_var.get(ax, "center")
    p = here if rng.random() < 0.9 else str(rng.choice(POSITIONS))
    q = _to_arg(rng, positions, where_var, ax) if ax in where_var else "center"
    dummy = "X" if rng.random() < 0.5 else ax  # a dummy axis name in the signature, mapped by `axis=`
    kw = {"axis": [(ax,)], "signature": sig.format(A=dummy, p=p, q=q)}
    if widths is not None:
        kw["padding_width"] = {ax: w for w in widths.values()}
    _padding_arg(rng, axes, kw)
    if rng.random() < 0.15:
        kw["pad_before_func"] = False
    return name, kw


def draw_call(rng, axes, positions, variables, where, metrics):
    fields = [v for v in variables if v.startswith("v")]
    var = str(rng.choice(fields))
    present = list(where[var])
    method = str(rng.choice(["diff", "interp", "min", "max", "diff", "interp", "cumsum", "cumsum", "derivative", "integrate",
                             "average", "cumint", "interp_like", "get_metric", "apply_as_grid_ufunc", "apply_as_grid_ufunc", "pad"]))
    kw, args = {}, []
    if metrics and rng.random() < 0.04:
        # the grid's metrics change under way: a registered name again (refused unless overwrite), another position, an
        # unknown variable, an unknown axis -- every later call of the case sees the same state on both grids
        names = [n for n in variables if n.startswith("m_")]
        key = _pick(rng, list(metrics)) if rng.random() < 0.85 else ("Q",)
        value = _pick(rng, names) if rng.random() < 0.85 else "no_such_variable"
        return "set_metrics", var, [key if rng.random() < 0.7 else list(key), value], {"overwrite": bool(rng.random() < 0.5)}
    if rng.random() < 0.06:
        # an operator CHAIN as a user writes it -- `(grid.diff(v, "X") - grid.diff(u, "Y")) / area`, `u * grid.interp(T, "X")`:
        # two operator results (or a result and a field) under + - * /, optionally over a metric variable; with `--fused` the
        # product defers and fuses what it can, and must agree with the reference's operator-by-operator chain
        def one():
            v = str(rng.choice(fields))
            ax = _pick(rng, list(where[v]) or axes)
            k = {}
            _padding_arg(rng, axes, k)
            return (_pick(rng, ["diff", "interp", "diff", "interp", "min", "max", "derivative"]), v, ax, k)
        left = one()
        right = one() if rng.random() < 0.75 else ("field", str(rng.choice(fields)), None, {})
        over = _pick(rng, [n for n in variables if n.startswith("m_")] or [None]) if rng.random() < 0.5 else None
        return "expr", left[1], [left, _pick(rng, ["add", "sub", "mul", "truediv"]), right, over], ({"_recast_left": True} if rng.random() < 0.3 else {})
    if method == "apply_as_grid_ufunc":
        name, kw = draw_user_ufunc(rng, axes, positions, where[var], present)
        return "apply_as_grid_ufunc:" + name, var, [], kw
    if method == "pad":
        widths = {ax: (int(rng.integers(0, 3)), int(rng.integers(0, 3))) for ax in (present if rng.random() < 0.9 else axes)
                  if rng.random() < 0.7}
        _padding_arg(rng, axes, kw)
        return "pad", var, [widths if (widths or rng.random() < 0.5) else None], kw
    with_metric = [ax for ax in present if (ax,) in metrics]
    if method in ("derivative", "integrate", "average", "cumint") and with_metric and rng.random() < 0.85:
        present = with_metric
    if method in ("diff", "interp", "min", "max"):
        axis = _axis_arg(rng, axes, present)
        args = [axis]
        _padding_arg(rng, axes, kw)
        r = rng.random()
        names = [axis] if isinstance(axis, str) else axis
        if r < 0.3 and len(names) == 1:
            kw["to"] = _to_arg(rng, positions, where[var], names[0])
        elif r < 0.45:
            kw["to"] = {ax: _to_arg(rng, positions, where[var], ax) for ax in names}
        if metrics and rng.random() < 0.25:
            kw["metric_weighted"] = _pick(rng, list(metrics)) if rng.random() < 0.7 else names[0]
    elif method == "cumsum":
        args = [_axis_arg(rng, axes, present)]
        _padding_arg(rng, axes, kw)
        names = [args[0]] if isinstance(args[0], str) else args[0]
        if rng.random() < 0.5:
            kw["to"] = {ax: _to_arg(rng, positions, where[var], ax) for ax in names} if (len(names) > 1 or rng.random() < 0.3) \
                else _to_arg(rng, positions, where[var], names[0])
        if rng.random() < 0.4:
            kw["reverse"] = bool(rng.random() < 0.7) if rng.random() < 0.7 else {ax: bool(rng.random() < 0.5) for ax in names}
        if metrics and rng.random() < 0.1:
            kw["metric_weighted"] = _pick(rng, list(metrics))
    elif method == "derivative":
        args = [str(rng.choice(present if rng.random() < 0.95 else axes))]
        _padding_arg(rng, axes, kw)
    elif method in ("integrate", "average"):
        args = [_axis_arg(rng, axes, present)]
        if rng.random() < 0.2:
            kw["skipna"] = bool(rng.random() < 0.5)
        if rng.random() < 0.1:
            kw["keep_attrs"] = True
    elif method == "cumint":
        args = [str(rng.choice(present))]
        _padding_arg(rng, axes, kw)
        if rng.random() < 0.4:
            kw["to"] = _to_arg(rng, positions, where[var], args[0])
        if rng.random() < 0.3:
            kw["reverse"] = True
    elif method == "interp_like":
        args = ["var:" + str(rng.choice(fields))]
        if rng.random() < 0.5:
            kw["padding"] = str(rng.choice(MODES))
        if rng.random() < 0.2:
            kw["fill_value"] = 1.5
    elif method == "get_metric":
        k = int(rng.integers(1, len(axes) + 1))
        args = [tuple(rng.choice(axes, size=k, replace=False).tolist())]
    if method not in ("get_metric", "interp_like") and rng.random() < 0.12:
        kw["_recast"] = True
    return method, var, args, kw


# ---- running one call on both grids ---------------------------------------------------------------------------------------
def _describe(res):
    return {"dims": tuple(res.dims), "name": res.name, "dtype": str(np.asarray(res.values).dtype), "coords": sorted(res.coords),
            "attrs": dict(res.attrs)}


def _call(grid, ds, method, var, args, kw, pad_function=None):
    args = [ds[a[4:]] if isinstance(a, str) and a.startswith("var:") else a for a in args]

    def operand(spec):
        if isinstance(spec, str) and spec.startswith("vec:"):
            _, ax, name = spec.split(":")
            return {ax: ds[name]}
        return ds[spec]

    if "other_component" in kw:
        kw = dict(kw, other_component=operand(kw["other_component"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            if method == "expr":
                import operator as _op

                def term(t, recast=False):
                    m, v, ax, k = t
                    da = ds[v]
                    if recast and "t" in da.dims:  # the two operands then disagree on `label` / `extra`: arithmetic drops those
                        da = da.assign_coords(label=("t", np.array([7, 8])), extra=("t", np.array([0.5, 1.5])))
                    return da if m == "field" else getattr(grid, m)(da, ax, **k)
                left, binop, right, over = args
                lt, rt = term(left, recast=bool(kw.get("_recast_left"))), term(right)
                if type(rt).__name__ == "LazyArray" and type(lt).__name__ != "LazyArray":
                    # an xarray object on the LEFT of a deferred result: xarray treats the stranger as an unlabelled array
                    # (operands meet by POSITION, xarray/core/variable.py `_broadcast_compat_data`) -- the documented way is
                    # to compute the deferred operand first; the chain then broadcasts by name as there
                    rt = rt.compute()
                res = getattr(_op, binop)(lt, rt)
                if over is not None:
                    res = res / ds[over]
                if type(res).__name__ == "LazyArray":
                    res = res.compute()
                return res, None
            if var.startswith("vec2:"):
                _, ux, vy = var.split(":")
                res = getattr(grid, method)({"X": ds[ux], "Y": ds[vy]}, **kw)
                return (res["X"], res["Y"]), None
            if method == "set_metrics":
                grid.set_metrics(*args, **kw)
                return None, None
            if method == "transform":
                spec = args[0]
                maker = type(ds[var])  # the container the dataset hands out (xarray stand-in / xgcm_amd's)
                target = spec[1] if spec[0] == "np" else maker(spec[2], dims=spec[1], name=spec[3])
                tkw = dict(kw)
                td = tkw.get("target_data")
                if isinstance(td, str):
                    tkw["target_data"] = ds[td.split(":")[1]] if td.startswith("var:") else ds[td.split(":")[1]].rename(None)
                res = grid.transform(ds[var], "Z", target, **tkw)
            elif var.startswith("vec:"):
                res = getattr(grid, method)(operand(var), *args, **kw)
            elif method.startswith("apply_as_grid_ufunc:"):
                res = grid.apply_as_grid_ufunc(USER_UFUNCS[method.split(":")[1]][0], ds[var], **kw)
            elif method == "pad":
                res = pad_function(ds[var], grid, args[0], **kw)
            else:
                da = ds[var]
                if kw.get("_recast"):  # coordinates the user changed on the INPUT (non-core ones must survive, GH #496)
                    kw = {k: v for k, v in kw.items() if k != "_recast"}
                    if "t" in da.dims:
                        da = da.assign_coords(t=np.array([100.0, 200.0]), label=("t", np.array([7, 8])), extra=("t", np.array([0.5, 1.5])))
                res = getattr(grid, method)(da, *args, **kw)
            if hasattr(res, "compute") and type(res).__name__ == "LazyArray":
                res = res.compute()  # (the deferred mode: the value is what is compared)
            return res, None
        except Exception as exc:  # noqa: BLE001 -- raising IS a behaviour to compare
            return None, exc


def compare(ref, ref_exc, got, got_exc):
    """None when both sides agree, else a short description of the first difference"""
    if got_exc is not None and "not part of the host build" in str(got_exc):
        return "OUTSIDE"  # the host build of the C ABI holds no token gathers (connected topologies): not a difference
    if (ref_exc is None) != (got_exc is None):
        return f"reference {'raised ' + type(ref_exc).__name__ + ': ' + str(ref_exc)[:100] if ref_exc else 'returned'}; " \
               f"xgcm_amd {'raised ' + type(got_exc).__name__ + ': ' + str(got_exc)[:100] if got_exc else 'returned'}"
    if ref_exc is not None:
        if ("more than 1 axis dimension" in str(got_exc) or "more than 1 axis dimension" in str(ref_exc)) \
                and isinstance(ref_exc, (ValueError, KeyError)) and isinstance(got_exc, (ValueError, KeyError)):
            # `field * metric` with the metric at ANOTHER position of the same axis (what default-shift interpolation of a
            # metric can produce) carries two dims of one axis; both stacks refuse to go on, the reference wherever its
            # container first trips over it (under the stand-in: numpy's "axes don't match array")
            return None
        if type(ref_exc).__name__ not in [c.__name__ for c in type(got_exc).__mro__]:  # (same class or a subclass of it)
            return f"exception type: reference {type(ref_exc).__name__} ({str(ref_exc)[:80]}), xgcm_amd {type(got_exc).__name__} ({str(got_exc)[:80]})"
        return None
    if ref is None or got is None:  # (a call made for its effect on the grid)
        return None if (ref is None and got is None) else f"reference {type(ref).__name__}, xgcm_amd {type(got).__name__}"
    if isinstance(ref, tuple) or isinstance(got, tuple):
        if not (isinstance(ref, tuple) and isinstance(got, tuple) and len(ref) == len(got)):
            return f"number of results: reference {type(ref).__name__}, xgcm_amd {type(got).__name__}"
        return next((d for d in (compare(r, None, g, None) for r, g in zip(ref, got)) if d is not None), None)
    a, b = _describe(ref), _describe(got)
    for key in ("dims", "name", "coords", "dtype", "attrs"):
        if a[key] != b[key]:
            return f"{key}: reference {a[key]!r}, xgcm_amd {b[key]!r}"
    x, y = np.asarray(ref.values), np.asarray(got.values)
    if x.shape != y.shape:
        return f"shape: reference {x.shape}, xgcm_amd {y.shape}"
    if not np.array_equal(x, y, equal_nan=x.dtype.kind == "f"):
        # (re-associated contiguous-axis scans / sums of the product, per dtype; float16 is a storage type of the product --
        # float32 lanes, one rounding on the way out: sums of float16 fields carry float32 partial sums)
        # (float16 sums: numpy rounds every partial sum to float16 -- n * 2^-11 * max|sum| of its own -- the lanes do not)
        tol = 1e-12 if x.dtype == np.float64 else (2e-6 if x.dtype == np.float32 else 3e-2)
        finite = np.abs(x[np.isfinite(x)].astype(float)) if x.dtype.kind == "f" else np.zeros(0)
        scale = max(1.0, float(finite.max())) if finite.size else 1.0  # (a re-associated scan errs by eps * its largest partial sum, also where it crosses zero)
        if not np.allclose(x, y, rtol=tol, atol=tol * scale, equal_nan=True):
            return f"values differ: max |d| = {np.nanmax(np.abs(x.astype(float) - y.astype(float))):.3e}"
    for c in a["coords"]:
        if not np.array_equal(np.asarray(ref.coords[c].values), np.asarray(got.coords[c].values)):
            return f"coordinate {c!r} differs"
        if tuple(ref.coords[c].dims) != tuple(got.coords[c].dims):
            return f"coordinate {c!r}: dims {ref.coords[c].dims} vs {got.coords[c].dims}"
    return None


def message_differs(ref_exc, got_exc):
    return ref_exc is not None and got_exc is not None and str(ref_exc)[:60] != str(got_exc)[:60]


def run(cases=100, seed=0, backend="oracle-double", calls_per_case=12, verbose=False, fused=False):
    xr, RefGrid, OurGrid = load_both(backend)
    from xgcm.padding import pad as ref_pad  # the reference's
    from xgcm_amd.padding import pad as our_pad
    stats = {"cases": 0, "calls": 0, "both_returned": 0, "both_raised": 0, "grid_errors_agreeing": 0}
    differences, messages = [], []
    for case in range(cases):
        ds, gkw, variables, calls = build_case(xr.Dataset, seed, case, calls_per_case)
        positions = gkw.get("coords", "autoparsed from the dataset's attributes")
        stats["cases"] += 1
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                rgrid, rexc = RefGrid(ds, **gkw), None
            except Exception as exc:  # noqa: BLE001
                rgrid, rexc = None, exc
            try:
                ogrid, oexc = OurGrid(ds, **dict(gkw, fuse=True) if fused else gkw), None
            except Exception as exc:  # noqa: BLE001
                ogrid, oexc = None, exc
        if rexc is not None or oexc is not None:
            if (rexc is None) != (oexc is None) or type(rexc).__name__ != type(oexc).__name__:
                differences.append({"case": case, "call": "Grid(...)", "kwargs": repr(gkw)[:300],
                                    "difference": f"reference {rexc!r:.120}; xgcm_amd {oexc!r:.120}"})
            else:
                stats["grid_errors_agreeing"] += 1
            continue
        for k, (method, var, args, kw) in enumerate(calls):
            ref, ref_exc = _call(rgrid, ds, method, var, args, kw, ref_pad)
            got, got_exc = _call(ogrid, ds, method, var, args, kw, our_pad)
            stats["calls"] += 1
            diff = compare(ref, ref_exc, got, got_exc)
            what = {"case": case, "k": k, "call": f"grid.{method}({var}{variables[var.split(':')[-1]][0]}, *{args}, **{kw})",
                    "grid": {kk: vv for kk, vv in gkw.items() if kk != "coords"}, "positions": positions}
            if diff == "OUTSIDE":
                stats["outside_the_host_build"] = stats.get("outside_the_host_build", 0) + 1
            elif diff is not None:
                differences.append(dict(what, difference=diff))
                if verbose:
                    print("DIFF", case, k, diff[:140])
            elif ref_exc is not None:
                stats["both_raised"] += 1
                why = type(ref_exc).__name__ + ": " + str(ref_exc)[:40]
                stats.setdefault("raised", {})[why] = stats.setdefault("raised", {}).get(why, 0) + 1
                if message_differs(ref_exc, got_exc):
                    messages.append(dict(what, reference=str(ref_exc)[:160], xgcm_amd=str(got_exc)[:160]))
            else:
                stats["both_returned"] += 1
    return stats, differences, messages


# ---- connected faces and the north fold --------------------------------------------------------------------------------
def draw_connected_grid(rng):
    """square faces joined by RANDOM links: every (face, axis, side) edge is left open or tied to another edge -- the same or
    another axis, the same or another face (itself included), reversed when both edges are on the same side (the rule of
    xgcm/grid.py:362-404)"""
    nf = int(rng.integers(1, 5))
    n = int(rng.integers(3, 6))
    edge_pos = "left" if rng.random() < 0.7 else "right"
    positions = {"X": {"center": "x", edge_pos: "xg"}, "Y": {"center": "y", edge_pos: "yg"}}
    off = 0.0 if edge_pos == "left" else 1.0
    coords = {"x": ("x", np.arange(n) + 0.5), "xg": ("xg", np.arange(n) + off), "y": ("y", np.arange(n) + 0.5),
              "yg": ("yg", np.arange(n) + off), "face": ("face", np.arange(nf)), "z": ("z", np.arange(2) * 1.0)}
    edges = [(f, ax, side) for f in range(nf) for ax in ("X", "Y") for side in (0, 1)]
    order = [edges[i] for i in rng.permutation(len(edges))]
    links = {f: {"X": [None, None], "Y": [None, None]} for f in range(nf)}
    free = list(order)
    while len(free) > 1:
        a = free.pop()
        if rng.random() < 0.3:
            continue  # an open edge: the boundary condition applies
        b = free.pop(int(rng.integers(0, len(free))))
        rev = bool(a[2] == b[2])
        links[a[0]][a[1]][a[2]] = (b[0], b[1], rev)
        links[b[0]][b[1]][b[2]] = (a[0], a[1], rev)
    conn = {"face": {f: {ax: tuple(v) for ax, v in d.items() if any(x is not None for x in v)} for f, d in links.items()}}
    conn["face"] = {f: d for f, d in conn["face"].items() if d}
    kw = {"coords": positions, "autoparse_metadata": False}
    if conn["face"]:
        kw["face_connections"] = conn
    if rng.random() < 0.85:
        kw["padding"] = str(rng.choice(MODES)) if rng.random() < 0.7 else {"X": str(rng.choice(MODES)), "Y": str(rng.choice(MODES))}
    if rng.random() < 0.4:
        kw["fill_value"] = float(rng.integers(-2, 3)) + 0.5
    variables, where = {}, {}
    lead = ("z", "face") if rng.random() < 0.7 else ("face",)
    for name, dims, pos in (("vc", ("y", "x"), {"X": "center", "Y": "center"}), ("vu", ("y", "xg"), {"X": edge_pos, "Y": "center"}),
                            ("vv", ("yg", "x"), {"X": "center", "Y": edge_pos}), ("vq", ("yg", "xg"), {"X": edge_pos, "Y": edge_pos})):
        shape = tuple({"z": 2, "face": nf}.get(d, n) for d in lead + dims)
        a = rng.standard_normal(shape)
        if rng.random() < 0.2:
            a.reshape(-1)[rng.integers(0, a.size, size=3)] = np.nan
        variables[name] = (lead + dims, a)
        where[name] = pos
    metrics = {}
    if rng.random() < 0.6:
        for ax, names in (("X", ("dx_c", "dx_u")), ("Y", ("dy_c", "dy_v"))):
            variables[names[0]] = (("face", "y", "x"), rng.random((nf, n, n)) + 0.5)
            variables[names[1]] = (("face",) + (("y", "xg") if ax == "X" else ("yg", "x")), rng.random((nf, n, n)) + 0.5)
            metrics[(ax,)] = list(names)
        kw["metrics"] = metrics
    sizes = {"x": n, "xg": n, "y": n, "yg": n, "face": nf, "z": 2}
    return ["X", "Y"], positions, sizes, coords, kw, variables, where, metrics, edge_pos


def draw_fold_grid(rng):
    ny, nx = int(rng.integers(3, 7)), 2 * int(rng.integers(2, 5))
    edge = "right" if rng.random() < 0.7 else "left"
    positions = {"X": {"center": "xh", edge: "xq"}, "Y": {"center": "yh", edge: "yq"}}
    off = 1.0 if edge == "right" else 0.0
    coords = {"xh": ("xh", np.arange(nx) + 0.5), "xq": ("xq", np.arange(nx) + off), "yh": ("yh", np.arange(ny) + 0.5),
              "yq": ("yq", np.arange(ny) + off), "z": ("z", np.arange(2) * 1.0)}
    r = rng.random()
    if r < 0.6:
        pivot = _pick(rng, ["center", "corner", "T", "F", "U", "V"])
    elif r < 0.9:
        pivot = {ax: _pick(rng, ["center", edge]) for ax in ("X", "Y") if rng.random() < 0.8} or {"Y": edge}
    else:
        pivot = "nowhere"  # not a pivot
    fold = {"fold": pivot}
    if rng.random() < 0.7:
        fold["south"] = str(rng.choice(MODES)) if rng.random() < 0.9 else "north"
    kw = {"coords": positions, "autoparse_metadata": False,
          "padding": {"X": "periodic" if rng.random() < 0.9 else "fill", "Y": fold}}
    if rng.random() < 0.4:
        kw["fill_value"] = {"Y": float(rng.integers(-2, 3)) + 0.5}
    variables, where = {}, {}
    for name, dims, pos in (("vt", ("yh", "xh"), {"X": "center", "Y": "center"}), ("vu", ("yh", "xq"), {"X": edge, "Y": "center"}),
                            ("vv", ("yq", "xh"), {"X": "center", "Y": edge}), ("vq", ("yq", "xq"), {"X": edge, "Y": edge})):
        lead = ("z",) if rng.random() < 0.6 else ()
        variables[name] = (lead + dims, rng.standard_normal(tuple({"z": 2, "yh": ny, "yq": ny}.get(d, nx) for d in lead + dims)))
        where[name] = pos
    sizes = {"xh": nx, "xq": nx, "yh": ny, "yq": ny, "z": 2}
    return ["X", "Y"], positions, sizes, coords, kw, variables, where, {}, edge


def draw_topology_call(rng, axes, positions, variables, where, metrics, edge_pos, vector_names):
    """single-axis calls only: with several padded axes the reference walks them in `set` order (xgcm/padding.py:481-487),
    which is no behaviour to pin"""
    fields = [v for v in variables if v.startswith("v")]
    var = _pick(rng, fields)
    ax = _pick(rng, axes)
    method = _pick(rng, ["diff", "interp", "diff", "interp", "min", "max", "cumsum", "pad", "vector", "vector", "derivative", "weighted"])
    kw = {}
    if rng.random() < 0.5:
        kw["padding"] = str(rng.choice(MODES))
        if rng.random() < 0.4:
            kw["fill_value"] = float(rng.integers(-2, 3)) + 0.25
    if method == "pad":
        return "pad", var, [{ax: (int(rng.integers(0, 3)), int(rng.integers(0, 3)))}], kw
    if method == "vector" and rng.random() < 0.25:
        # both components at once (the deprecated pair of methods the reference keeps: xgcm/grid.py:1420-1500)
        ux, vy = vector_names
        if rng.random() < 0.15:
            kw["to"] = _pick(rng, ["center", edge_pos])
        return _pick(rng, ["interp_2d_vector", "diff_2d_vector"]), f"vec2:{ux}:{vy}", [], kw
    if method == "vector":  # a vector component with its partner: halos across rotated / reversed links take the partner
        ux, vy = vector_names
        comp, other = (("X", ux), ("Y", vy)) if rng.random() < 0.5 else (("Y", vy), ("X", ux))
        kw["other_component"] = f"vec:{other[0]}:{other[1]}"
        return _pick(rng, ["diff", "interp"]), f"vec:{comp[0]}:{comp[1]}", [ax], kw
    if method == "cumsum":
        if rng.random() < 0.5:
            kw["reverse"] = True
        if rng.random() < 0.4:
            kw["to"] = _to_arg(rng, positions, where[var], ax)
        return "cumsum", var, [ax], kw
    if method == "derivative":
        return "derivative", var, [ax], kw
    if method == "weighted":
        kw["metric_weighted"] = (ax,) if rng.random() < 0.8 else ("X", "Y")
        return _pick(rng, ["diff", "interp"]), var, [ax], kw
    if rng.random() < 0.3:
        kw["to"] = _to_arg(rng, positions, where[var], ax)
    return method, var, [ax], kw


# ---- the vertical transform ----------------------------------------------------------------------------------------------
def _theta(rng, shape, kind):
    """columns along the LAST dim: increasing / decreasing / a random walk (non-monotonic) / with duplicates / NaN tails"""
    steps = rng.random(shape) + 0.05
    if kind == "walk":
        steps = rng.standard_normal(shape)
    col = np.cumsum(steps, axis=-1) + 1.0
    if kind == "decreasing":
        col = col[..., ::-1].copy()
    if kind == "duplicates":
        col[..., 2] = col[..., 1]
    if kind == "nan_tail":
        col[..., -2:] = np.nan
    return col


def draw_transform_case(rng, make_dataset, calls_per_case):
    nz, nx = int(rng.integers(4, 9)), int(rng.integers(1, 4))
    extra = _pick(rng, ["outer", "outer", "outer", "left", "right", "inner"])
    positions = {"Z": {"center": "z", extra: "zg"}}
    nzg = nz + LENGTH[extra]
    coords = {"z": ("z", np.arange(nz) + 0.5), "zg": ("zg", np.arange(nzg) * 1.0 + (0.0 if extra in ("outer", "left") else 1.0)),
              "x": ("x", np.arange(nx) * 10.0), "t": ("t", np.array([0.0, 1.0]))}
    gkw = {"coords": positions, "autoparse_metadata": False}
    r = rng.random()
    if r < 0.3:
        gkw["padding"] = _pick(rng, ["fill", "extend", "fill", "periodic"])
    elif r < 0.4:
        gkw["padding"] = {"Z": _pick(rng, ["fill", "extend"])}
    lead = _pick(rng, [(), ("x",), ("x",), ("t", "x")])
    if rng.random() < 0.2 and lead:
        order = lambda dims: tuple(rng.permutation(list(dims)))  # noqa: E731
    else:
        order = lambda dims: tuple(dims)  # noqa: E731
    shape_of = lambda dims: tuple({"z": nz, "zg": nzg, "x": nx, "t": 2}[d] for d in dims)  # noqa: E731
    variables = {}

    def put(name, dims, values_last_axis_fn):
        canon = tuple(lead) + (dims[-1],)
        vals = values_last_axis_fn(shape_of(canon))
        final = order(canon)
        variables[name] = (final, np.transpose(vals, [canon.index(d) for d in final]).copy())

    kind = _pick(rng, ["increasing", "increasing", "decreasing", "walk", "duplicates", "nan_tail"])
    put("vd", ("z",), lambda sh: rng.standard_normal(sh) if rng.random() < 0.8 else rng.standard_normal(sh).astype(np.float32))
    if rng.random() < 0.25:
        variables["vd"][1].reshape(-1)[rng.integers(0, variables["vd"][1].size, size=2)] = np.nan
    put("vth", ("z",), lambda sh: _theta(rng, sh, kind))
    put("vthg", ("zg",), lambda sh: _theta(rng, sh, kind))
    variables["vth1"] = (("z",), _theta(rng, (nz,), kind))  # a 1-D profile: fewer dims than the data
    variables["vother"] = (("t", "q", "z") if rng.random() < 0.5 else ("q", "z"), None)
    q = variables.pop("vother")[0]
    variables["vthq"] = (q, _theta(rng, tuple({"t": 2, "q": 2, "z": nz}[d] for d in q), "increasing"))  # a dim the data lacks
    coords["q"] = ("q", np.arange(2) * 1.0)
    ds = make_dataset({k: v for k, v in variables.items()}, coords)
    lo, hi = float(np.nanmin(variables["vth"][1])), float(np.nanmax(variables["vth"][1]))
    calls = []
    for _ in range(calls_per_case):
        # (no unknown method names: the reference crashes on one -- UnboundLocalError, `out` never assigned,
        # xgcm/transform.py:515 -- where xgcm_amd raises a ValueError that names the methods; a stated deviation)
        method = _pick(rng, ["linear", "linear", "conservative", "conservative", "log"])
        nlev = int(rng.integers(2, 7))
        levels = np.sort(rng.uniform(lo - 0.5, hi + 0.5, nlev))
        r = rng.random()
        # (decreasing conservative bins only on 1-D data: on N-D data the reference reverses axis 0 of its result -- time,
        # here -- instead of the bin axis, xgcm/transform.py:190-192; xgcm_amd reverses the bins: a stated deviation)
        if r < 0.15 and (method != "conservative" or not lead):
            levels = levels[::-1].copy()
        elif 0.15 <= r < 0.22 and (method != "conservative" or not lead):
            levels = rng.permutation(levels)
        if rng.random() < 0.1:
            levels = levels.astype(np.float32)
        if method == "log":
            # (logarithms of positive float64 numbers only: a column that is all NaN after `np.log` and the float32 rounding
            # of `log` -- once in double here, DESIGN section 2 -- are stated deviations)
            if kind == "walk" or variables["vd"][1].dtype != np.float64:
                method = "linear"
            levels = (np.abs(levels) + 0.25).astype(np.float64)
        r = rng.random()
        if r < 0.35:
            target = ("np", levels)
        elif r < 0.8:
            target = ("da", (_pick(rng, ["lev", "vth", "z"]),), levels, _pick(rng, [None, "lev", "theta"]))
        else:  # a target that varies in x: needs `target_dim`
            target = ("da", ("x", "lev"), np.sort(rng.uniform(lo, hi, (nx, nlev)), axis=-1), None)
        kw = {"method": method}
        td = _pick(rng, [None, None, "var:vth", "var:vth", "var:vthg", "var:vthg", "var:vth1", "var:vthq", "unnamed:vth"])
        if td is not None:
            kw["target_data"] = td
        if rng.random() < 0.3:
            kw["mask_edges"] = bool(rng.random() < 0.5)
        if rng.random() < 0.15:
            kw["bypass_checks"] = True
        if rng.random() < 0.15:
            kw["suffix"] = "_on_levels"
        if target[0] == "da" and len(target[1]) == 2 or rng.random() < 0.1:
            kw["target_dim"] = _pick(rng, ["lev", "lev", "lev", "nowhere"])
        calls.append(("transform", "vd", [target], kw))
    return ds, gkw, variables, calls


def build_case(make_dataset, seed, case, calls_per_case=12):
    """the seeded inputs of one case WITHOUT the reference: (dataset, grid kwargs, variables, [(method, var, args, kwargs)])"""
    rng = np.random.default_rng([seed, case])
    kind = rng.random()
    if kind > 0.88:  # the vertical transform
        return draw_transform_case(rng, make_dataset, calls_per_case)
    if kind < 0.3:  # a complex topology: connected faces (0.2) or a north fold (0.1)
        draw = draw_connected_grid if kind < 0.2 else draw_fold_grid
        axes, positions, sizes, coords, gkw, variables, where, metrics, edge_pos = draw(rng)
        ds = make_dataset({k: v for k, v in variables.items()}, coords)
        vector_names = ("vu", "vv")
        calls = [draw_topology_call(rng, axes, positions, variables, where, metrics, edge_pos, vector_names) for _ in range(calls_per_case)]
        return ds, gkw, variables, calls
    axes, positions, sizes, coords, gkw = draw_grid(rng)
    long = gkw.pop("_long", None)
    variables, where = draw_variables(rng, axes, positions, sizes, long)
    metrics = draw_metrics(rng, axes, positions, sizes, variables, long)
    if metrics:
        gkw["metrics"] = metrics
    ds = make_dataset({k: v for k, v in variables.items()}, coords)
    calls = [draw_call(rng, axes, positions, variables, where, metrics) for _ in range(calls_per_case)]
    gkw.pop("_autoparsed", None)
    return ds, gkw, variables, calls


SAMPLE_ABOVE = 20000  # results with more cells are stored as every k-th cell (shape kept): the fixture stays a few MB


def sample_step(size):
    return 1 if size <= SAMPLE_ABOVE else -(-size // 4096)


def _pack(res):
    outs = res if isinstance(res, tuple) else (res,)
    return [{"dims": list(o.dims), "name": o.name, "coords": {c: list(o.coords[c].dims) for c in sorted(o.coords)},
             "shape": [int(n) for n in np.asarray(o.values).shape]} for o in outs]


def record(cases, seed, out_prefix):
    """The reference's answers to the generator's calls as a fixture: `<out_prefix>.json` (per call: raised type, or dims /
    name / coordinate names of every result) + `<out_prefix>.npz` (values of results and of their coordinates).  The
    INPUTS are not stored: `build_case(seed, case)` regenerates them anywhere numpy is."""
    xr, RefGrid, _ = load_both()
    from xgcm.padding import pad as ref_pad

    meta = {"seed": seed, "cases": cases, "calls_per_case": 12, "what": "expected outcomes of oracle/fuzz_against_reference.py's "
            "seeded calls, produced by the reference's own Grid (over the xarray stand-in); inputs are regenerated from the seed",
            "outcomes": []}
    arrays = {}
    for case in range(cases):
        ds, gkw, variables, calls = build_case(xr.Dataset, seed, case)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                grid = RefGrid(ds, **gkw)
            except Exception as exc:  # noqa: BLE001
                meta["outcomes"].append({"grid_raises": type(exc).__name__})
                continue
        per_call = []
        for k, (method, var, args, kw) in enumerate(calls):
            res, exc = _call(grid, ds, method, var, args, kw, ref_pad)
            if exc is not None:
                per_call.append({"raises": type(exc).__name__, "message": str(exc)[:80]})
                continue
            if res is None:
                per_call.append({"results": []})
                continue
            outs = res if isinstance(res, tuple) else (res,)
            per_call.append({"results": _pack(res)})
            for j, o in enumerate(outs):
                vals = np.asarray(o.values)
                step = sample_step(vals.size)
                arrays[f"{case}/{k}/{j}"] = vals if step == 1 else np.ascontiguousarray(vals.reshape(-1)[::step])
                for c in o.coords:
                    arrays[f"{case}/{k}/{j}/coord/{c}"] = np.asarray(o.coords[c].values)
        meta["outcomes"].append({"calls": per_call})
    with open(out_prefix + ".json", "w") as f:
        json.dump(meta, f, indent=0)
    np.savez_compressed(out_prefix + ".npz", **arrays)
    n = sum(len(o.get("calls", [])) for o in meta["outcomes"])
    print(f"{cases} cases, {n} calls ({sum('results' in c for o in meta['outcomes'] for c in o.get('calls', []))} results) -> {out_prefix}.json/.npz")


def record_stable(cases, seed, out_prefix, hash_seeds=tuple(range(1, 15))):
    """`record` under several PYTHONHASHSEEDs: where the reference's answer depends on the iteration order of a set of
    strings (`iterate_axis_combinations` walks `combinations(frozenset(axes))`: WHICH metrics get multiplied, and in which
    dim order, for a multi-axis metric) the call is marked `hash_seed_dependent` and carries no expectation -- such
    behaviour is excluded from every fixture of this repository.  Within one process both stacks
    iterate the same sets the same way: the live comparison (`run`) covers those calls."""
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="xgcm_fuzz_record_")
    metas, npzs = [], []
    for hs in hash_seeds:
        prefix = os.path.join(tmp, f"h{hs}")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--cases", str(cases), "--seed", str(seed), "--record-raw", prefix],
                       env=dict(os.environ, PYTHONHASHSEED=str(hs)), check=True, capture_output=True)
        with open(prefix + ".json") as f:
            metas.append(json.load(f))
        npzs.append(np.load(prefix + ".npz"))
    meta, keep, unstable = metas[0], {}, 0
    for case, outcome in enumerate(meta["outcomes"]):
        for k, call in enumerate(outcome.get("calls", [])):
            same = all(m["outcomes"][case]["calls"][k] == call for m in metas[1:])
            keys = [key for key in npzs[0].files if key.startswith(f"{case}/{k}/")]
            if same:
                same = all(set(keys) == {key for key in z.files if key.startswith(f"{case}/{k}/")} and
                           all(np.array_equal(npzs[0][key], z[key], equal_nan=npzs[0][key].dtype.kind == "f") for key in keys)
                           for z in npzs[1:])
            if same:
                keep.update({key: npzs[0][key] for key in keys})
            else:
                outcome["calls"][k] = {"hash_seed_dependent": True}
                unstable += 1
    meta["hash_seeds_compared"] = list(hash_seeds)
    with open(out_prefix + ".json", "w") as f:
        json.dump(meta, f, indent=0)
    np.savez_compressed(out_prefix + ".npz", **keep)
    n = sum(len(o.get("calls", [])) for o in meta["outcomes"])
    print(f"{cases} cases, {n} calls, {unstable} of them hash-seed dependent and left out -> {out_prefix}.json/.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="oracle-double", choices=["oracle-double", "host-abi", "hip"])
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--fused", action="store_true", help="xgcm_amd.Grid(fuse=True): deferred results, computed for the comparison")
    ap.add_argument("--record-raw", metavar="PREFIX", help=argparse.SUPPRESS)
    ap.add_argument("--record", metavar="PREFIX", help="write the reference's answers as a fixture (tests/golden/fuzz_reference)")
    args = ap.parse_args()
    if args.record_raw:
        record(args.cases, args.seed, args.record_raw)
        return
    if args.record:
        record_stable(args.cases, args.seed, args.record)
        return
    stats, differences, messages = run(args.cases, args.seed, args.backend, verbose=args.verbose, fused=args.fused)
    print(json.dumps(stats))
    by = {}
    for d in differences:
        by.setdefault(d["difference"][:90], []).append(d)
    for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(len(v), "x", k, "\n      e.g.", v[0]["call"][:200], "| grid", repr(v[0].get("grid"))[:200])
    print(len(messages), "agreeing exceptions with different wording")
    by = {}
    for m in messages:
        by.setdefault((m["reference"][:70], m["xgcm_amd"][:70]), []).append(m)
    for (a, b), v in sorted(by.items(), key=lambda kv: -len(kv[1]))[:15]:
        print(len(v), "x", "\n   ref:", a, "\n   own:", b)
    sys.exit(1 if differences else 0)


if __name__ == "__main__":
    main()
