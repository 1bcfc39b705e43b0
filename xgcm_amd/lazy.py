"""Deferred operator results: the operator CHAIN stays the interface, the fused kernels become its execution.

The reference applies one grid ufunc per axis and materialises every intermediate (`xgcm/grid.py:797-832`; the TODO at
`:797-799` and `docs/grid_ufuncs.md:27` name exactly that as the cost).  A drop-in user therefore writes BASELINE
configs[4] as `(grid.diff(v, "X") - grid.diff(u, "Y")) / area` -- four kernels and 56 B per cell where the fused
`xg_vorticity` moves 24.  With `Grid(..., fuse=True)` or inside `with grid.fused():` the built-in 1-D operators return a
`LazyArray`: a `DataArray` whose dims / coords / name / shape / dtype are known at once and whose `.data` is computed on
first use.  `+ - * /` between such results and ordinary arrays build a small expression; when a value is needed the
expression is matched against the kernels that already exist and otherwise evaluated node by node EXACTLY as the eager
path would (same calls, same order):

    (diff(v, X) - diff(u, Y)) [/ area]       -> xg_vorticity        (one launch)
    (diff(u, X) + diff(v, Y)) [/ area]       -> xg_divergence
    diff(a, X), diff(a, Y)  (or derivative)  -> xg_gradient         (siblings on one source: both in one launch)
    u * interp(T, X), v * interp(T, Y)       -> xg_flux             (siblings)
    op(op(a, X), Y)                          -> xg_stencil2d        (the two axes in one pass)
    op(a * m, X) / n                         -> xg_stencil1d with m_in / m_out (the metric rides in the stencil's launch)

Every rule is bit-identical to the chain it replaces: the kernels evaluate the same IEEE operations in the same order
(tests/test_lazy_fusion.py compares with the eager chain on every backend; the fused kernels themselves are compared
with the chain at full size in tests/test_gpu_fullsize.py).  A rule applies only when dtypes are uniform float32 /
float64 (mixed precision, float16 and integers keep numpy's step-by-step promotion: node by node), layouts are the
kernels' ((Y, X) last) and every sub-expression is still unevaluated; anything else falls back to the eager sequence.

What deferral changes, and the switch is opt-in because of it: inputs are read when the value is first used (mutating
an input array between the call and the use is seen); a missing boundary condition on an ordinary axis still raises
at the call, topology errors of connected axes are checked at the call by building (not applying) the halo map.
"""

from __future__ import annotations

import threading
import weakref
from collections import OrderedDict
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np

from . import device as _dev
from . import dtypes as _dt
from .labeled import DataArray, _aligned_view, _is_tensor

try:  # torch is plumbing for device memory only
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore

# how values were produced since the last reset (tests and tools assert that a fused kernel really ran)
STATS: Dict[str, int] = {}


def _count(rule: str) -> None:
    STATS[rule] = STATS.get(rule, 0) + 1


def reset_stats() -> None:
    STATS.clear()


_FUSABLE_FLOATS = (_dt.FLOAT32, _dt.FLOAT64)


class Node:
    """one deferred operation; `value` is the raw result array (numpy or HBM tensor) once computed"""

    __slots__ = ("value", "__weakref__")

    def get(self):
        if self.value is None:
            self.set(self._compute())
        return self.value

    def set(self, value) -> None:
        """the node's value; its operands are let go (a result must not keep its inputs alive any longer than the eager
        call would have)"""
        self.value = value
        self._release()

    def _release(self) -> None:
        pass

    def _compute(self):  # pragma: no cover
        raise NotImplementedError


def _unforced(x, cls=None):
    """the node of a LazyArray whose value has not been computed (optionally of a given class), else None"""
    if isinstance(x, LazyArray) and x._node.value is None and (cls is None or isinstance(x._node, cls)):
        return x._node
    return None


def plain(x):
    """a LazyArray as an ordinary DataArray (computing it), dict components likewise; anything else unchanged"""
    if isinstance(x, LazyArray):
        return x._plain()
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    return x


def like(array):
    """what `Grid._resident(metric, like)` needs to know of an array -- is it in HBM? -- without computing a deferred one"""
    return array._like if isinstance(array, LazyArray) else array.data


def _source_key(da) -> Tuple:
    """identity of an operand's data: the array object itself, or the node of a deferred one"""
    if isinstance(da, LazyArray):
        return ("node", id(da._node))
    return ("data", id(da.data))


def _dtype_of(x) -> np.dtype:
    return _dt.np_dtype(x.dtype if isinstance(x, LazyArray) else x.data)


def _is_host(x) -> bool:
    return x._host if isinstance(x, LazyArray) else not _is_tensor(x.data)


class LazyArray(DataArray):
    """A DataArray whose data is a deferred operation (see the module docstring).  Metadata never computes; `.data`,
    `.values` and everything built on them do, once, and the value is shared by every relabelled copy."""

    __slots__ = ("_node", "_lshape", "_ldtype", "_host", "_xr")

    def __init__(self, node: Node, dims, shape, dtype, host: bool, coords=None, name=None, attrs=None):
        self._node = node
        self._lshape = tuple(int(s) for s in shape)
        self._ldtype = np.dtype(dtype)
        self._host = bool(host)
        self._xr = False  # results of xarray inputs hand an xarray.DataArray over when computed (set by Grid, kept by + - * /)
        self.dims = tuple(dims)
        self.name = name
        self.attrs = dict(attrs) if attrs else {}
        self.coords = OrderedDict()
        if coords:
            self._set_coords(coords)

    # ---- the one deferred attribute ------------------------------------------------------
    @property
    def data(self):
        return self._node.get()

    @data.setter
    def data(self, value):  # pragma: no cover
        raise AttributeError("the data of a deferred result is computed, not assigned")

    @property
    def chunks(self):
        return None  # (asking must not force the value: DataArray.chunks looks at `.data`)

    @property
    def is_deferred(self) -> bool:
        return self._node.value is None

    def compute(self):
        """the same labelled array with its data evaluated: an ordinary DataArray -- an `xarray.DataArray` when the
        chain started from xarray objects, which is what `.compute()` of a dask-backed result gives there"""
        out = self._plain()
        return _labeled.to_xarray(out) if self._xr else out

    load = compute

    def _plain(self) -> DataArray:
        out = DataArray.__new__(DataArray)
        out.data = self.data
        out.dims, out.name, out.attrs, out.coords = self.dims, self.name, dict(self.attrs), OrderedDict(self.coords)
        return out

    # ---- metadata without evaluation -----------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        return self._lshape

    @property
    def dtype(self):
        return self._ldtype

    @property
    def is_device(self) -> bool:
        return not self._host

    @property
    def _like(self):
        if self._host or torch is None:
            return None
        return torch.empty(0, device="cuda")

    def _replace(self, data=None, dims=None, coords=None, name="__keep__"):
        if data is not None or (dims is not None and len(tuple(dims)) != len(self.dims)):
            return self._plain()._replace(data=data, dims=dims, coords=coords, name=name)
        out = LazyArray.__new__(LazyArray)
        out._node, out._lshape, out._ldtype, out._host = self._node, self._lshape, self._ldtype, self._host
        out._xr = self._xr
        out.dims = self.dims if dims is None else tuple(dims)  # (a rename: same cells, same order)
        out.name = self.name if name == "__keep__" else name
        out.attrs = dict(self.attrs)
        out.coords = OrderedDict(self.coords if coords is None else coords)
        return out

    def copy(self, deep: bool = False, data=None):
        if deep or data is not None:
            return self._plain().copy(deep=deep, data=data)
        return self._replace()

    def to_xarray(self):
        from .labeled import to_xarray

        return to_xarray(self)

    # ---- arithmetic stays deferred while an operand is ------------------------------------
    def _binary(self, other, op: str, reflexive: bool = False, dims_order: Optional[Sequence[str]] = None):
        xr_out = self._xr or getattr(other, "_xr", False)
        if _labeled.is_xarray(other):
            other, xr_out = _labeled.from_xarray(other), True
        res = defer_binary(self, other, op, reflexive, dims_order)
        if res is not None:
            res._xr = xr_out
            return res
        res = DataArray._binary(self._plain(), plain(other), op, reflexive, dims_order)
        return _labeled.to_xarray(res) if xr_out else res

    def __abs__(self):
        out = DataArray.__abs__(self._plain())
        return _labeled.to_xarray(out) if self._xr else out

    # ---- a deferred result of xarray inputs answers xarray's methods with xarray objects ----
    def isel(self, *a, **k):
        return self.compute().isel(*a, **k) if self._xr else DataArray.isel(self, *a, **k)

    def transpose(self, *a, **k):
        return self.compute().transpose(*a, **k) if self._xr else DataArray.transpose(self, *a, **k)

    @property
    def T(self):
        return self.compute().T if self._xr else DataArray.T.fget(self)

    def rename(self, *a, **k):
        return self.compute().rename(*a, **k) if self._xr else DataArray.rename(self, *a, **k)

    def reset_coords(self, *a, **k):
        return self.compute().reset_coords(*a, **k) if self._xr else DataArray.reset_coords(self, *a, **k)

    def drop_vars(self, *a, **k):
        return self.compute().drop_vars(*a, **k) if self._xr else DataArray.drop_vars(self, *a, **k)

    def assign_coords(self, *a, **k):
        return self.compute().assign_coords(*a, **k) if self._xr else DataArray.assign_coords(self, *a, **k)

    def to_dataset(self, *a, **k):
        return self.compute().to_dataset(*a, **k) if self._xr else DataArray.to_dataset(self, *a, **k)

    def sum(self, *a, **k):
        return self.compute().sum(*a, **k) if self._xr else DataArray.sum(self, *a, **k)

    def cumsum(self, *a, **k):
        return self.compute().cumsum(*a, **k) if self._xr else DataArray.cumsum(self, *a, **k)

    def __getattr__(self, key: str):
        """What this class does not have, the computed xarray object does (`.plot`, `.sel`, `.isel`, `.where` ...):
        a deferred result of xarray inputs answers like the `xarray.DataArray` it stands for, as a dask-backed one does."""
        try:
            return DataArray.__getattr__(self, key)
        except AttributeError:
            if key.startswith("_"):
                raise
            try:
                from_xarray = object.__getattribute__(self, "_xr")
            except AttributeError:
                from_xarray = False
            if not from_xarray:
                raise
        return getattr(self.compute(), key)

    def __repr__(self) -> str:
        state = "deferred" if self.is_deferred else ("HBM" if self.is_device else "host")
        return f"<xgcm_amd.LazyArray {self.name!r} {dict(self.sizes)} [{state}] coords={list(self.coords)}>"


# ==============================================================================================
# stencil nodes: one axis of Grid.diff / interp / min / max / derivative
# ==============================================================================================
_PENDING = "_lazy_pending"
_PENDING_LOCK = threading.Lock()  # one Grid may be used from several threads: its set of unevaluated nodes is shared


def _pending_add(grid, node) -> None:
    with _PENDING_LOCK:
        ws = grid.__dict__.get(_PENDING)
        if ws is None:
            ws = grid.__dict__[_PENDING] = weakref.WeakSet()
        ws.add(node)


def _pending_list(grid) -> list:
    with _PENDING_LOCK:
        ws = grid.__dict__.get(_PENDING)
        return list(ws) if ws is not None else []


class StencilNode(Node):
    __slots__ = ("grid", "funcname", "ufunc", "sig", "arg", "source", "ax_name", "other_component", "remaining", "m_in",
                 "m_out", "in_dim", "out_dim", "out_dims", "lo", "hi", "bc", "fv", "complex")

    def __init__(self, **kw):
        self.value = None
        for k, v in kw.items():
            setattr(self, k, v)

    def _release(self) -> None:
        self.arg = self.source = self.other_component = self.m_in = self.m_out = None

    # -- the eager call, optionally with metrics that a surrounding expression contributes ------------
    def run(self, extra_m_out: Optional[DataArray] = None):
        """exactly the call `_1d_grid_ufunc_dispatch` makes for this step; an unevaluated `a * m` argument rides along as
        the input metric, `extra_m_out` (from an enclosing `/ n`) as the output metric -- both are what `metric_weighted`
        / `derivative` hand to the same kernel"""
        arg, m_in = self.arg, self.m_in
        picked = self._product_as_metric() if m_in is None else None
        if picked is not None:
            arg, m_in = picked
            _count("stencil_m_in")
        m_out = self.m_out
        if extra_m_out is not None:
            assert m_out is None
            m_out = extra_m_out
            _count("stencil_m_out")
        res = self.ufunc(self.grid, plain(arg), axis=[(self.ax_name,)], other_component=plain(self.other_component),
                         metric_in=m_in, metric_out=m_out, **self.remaining)
        return res.data

    def _product_as_metric(self):
        """`op(a * m, X)` with the product still unevaluated, `a` and `m` of one float dtype, `m` broadcast along dims of
        `a`: (a, m) -- the kernel forms the same rounded products"""
        node = _unforced(self.arg, BinaryNode)
        if node is None or node.op != "mul" or self.other_component is not None:
            return None
        a, m = (node.b, node.a) if node.reflexive else (node.a, node.b)
        if not (isinstance(a, DataArray) and isinstance(m, DataArray)):
            return None
        if isinstance(m, LazyArray) or isinstance(a, LazyArray):
            if _unforced(a) is not None or _unforced(m) is not None:
                return None
        if not set(m.dims) <= set(a.dims):  # the field is the operand that has every dim
            a, m = m, a
            if not set(m.dims) <= set(a.dims):
                return None
        if tuple(a.dims) != tuple(self.arg.dims) or _dtype_of(a) != _dtype_of(m) or _dtype_of(a) not in _FUSABLE_FLOATS:
            return None
        if self.complex:  # halos of the PRODUCT through the topology: the explicit product pass keeps them simple
            return None
        return plain(a), self.grid._resident(plain(m), plain(a).data)

    def _compute(self):
        got = _try_gradient(self)
        if got is not None:
            return got
        got = _try_two_axes(self)
        if got is not None:
            return got
        _count("stencil_eager")
        return self.run()

    # -- what the pattern rules ask of a node ------------------------------------------------------------
    def plain_diff(self, from_pos: str, to_pos: str) -> bool:
        return (self.funcname == "diff" and self.m_in is None and self.m_out is None
                and (self.ufunc.from_pos, self.ufunc.to_pos) == (from_pos, to_pos) and (self.lo + self.hi) == 1)

    def halo(self, widths):
        """(boundary mode, fill value, pre-gathered halo slab or None) of this node's axis, as the node itself pads"""
        from .padding import halo_cells

        if self.complex:
            h = halo_cells(plain(self.arg), self.grid, self.ax_name, widths, padding=self.remaining.get("padding"),
                           fill_value=self.remaining.get("fill_value"), other_component=plain(self.other_component))
            return "halo", 0.0, h.data
        return self.bc, float(self.fv or 0.0), None


def defer_stencil(grid, funcname, ufunc, sig, arg, ax_name, other_component, m_in, m_out, remaining, out_dims):
    """A LazyArray for one step of `_1d_grid_ufunc_dispatch`, or None when the step must run now (so that it raises what
    the eager call raises, or because nothing about it can be fused later)."""
    from . import gridops
    from .grid_ufunc import _maybe_unpack_vector_component, _reattach_coords
    from .padding import halo_cells, no_boundary_error  # noqa: F401

    if funcname not in ("diff", "interp", "min", "max"):
        return None
    kwargs = dict(remaining)
    if set(kwargs) - {"padding", "fill_value"}:
        return None
    call_kw = dict(kwargs, other_component=other_component, metric_in=m_in, metric_out=m_out)
    if not ufunc._fusable(grid, (arg,), [(ax_name,)], call_kw):
        return None
    try:  # the argument checks of the eager call (gridops._fused -> _check_data_input): a malformed argument raises NOW
        from .grid_ufunc import _check_data_input

        _check_data_input(arg, grid)
        _check_data_input(other_component, grid)
    except Exception:  # noqa: BLE001 -- the eager call the caller falls back to raises it
        return None
    da = _maybe_unpack_vector_component(arg)
    dt = _dtype_of(da)
    if dt not in _FUSABLE_FLOATS:
        return None  # integers, float16: numpy's step-by-step dtypes, computed now
    for m in (m_in, m_out):
        if m is not None and _dtype_of(m) != dt:
            return None
    axis = grid.axes[ax_name]
    try:
        in_dim, out_dim = axis.coords[ufunc.from_pos], axis.coords[ufunc.to_pos]
    except KeyError:
        return None
    if in_dim not in da.dims:
        return None
    try:  # a metric that does not fit the array (ill-formed products of odd metric positions): the eager call raises now
        for m, dims in ((m_in, da.dims), (m_out, tuple(out_dims))):
            if m is not None:
                _labeled._aligned_view(m, dims)
    except Exception:  # noqa: BLE001
        return None
    lo, hi = next(iter(ufunc.padding_width.values())) if ufunc.padding_width else (0, 0)
    complex_ = gridops.complex_topology(grid, ax_name)
    bc = fv = None
    if complex_ and (lo or hi):
        # the topology's own checks (unconnected edges without a boundary condition, vector components without their
        # partner ...) run now: the halo map is BUILT -- and cached for the evaluation -- but nothing is moved
        from .padding import pad

        pad(arg, grid, {ax_name: (lo, hi)}, padding=kwargs.get("padding"), fill_value=kwargs.get("fill_value"),
            other_component=other_component, _halo_only=ax_name, _dry=True)
    else:
        bc = grid._complete_user_kwargs_using_axis_defaults(kwargs.get("padding"), "padding")[ax_name]
        fv = grid._complete_user_kwargs_using_axis_defaults(kwargs.get("fill_value"), "fill_value")[ax_name]
        if (lo or hi) and not isinstance(bc, str):
            return None  # no boundary condition: the eager call raises the reference's error now
        faces = getattr(grid, "_face_connections", None)
        if faces is not None and (lo or hi) and grid._facedim in da.dims and \
                any(i not in faces[grid._facedim] for i in range(da.sizes[grid._facedim])):
            return None  # a face the connections leave out, on an axis no link touches: the eager call's KeyError, now
        if not (lo or hi):
            bc = None
    node = StencilNode(grid=grid, funcname=funcname, ufunc=ufunc, sig=sig, arg=arg, source=da, ax_name=ax_name,
                       other_component=other_component, remaining=kwargs, m_in=m_in, m_out=m_out, in_dim=in_dim,
                       out_dim=out_dim, out_dims=tuple(out_dims), lo=int(lo), hi=int(hi), bc=bc, fv=fv, complex=complex_)
    shape = tuple(n + lo + hi - 1 if d == in_dim else n for d, n in zip(da.dims, da.shape))
    name = da.name
    if complex_ and (lo or hi) and m_in is None:
        # (on connected faces the reference's pad may leave the PARTNER's name on a vector component: face_concat_name)
        from .padding import face_concat_name

        partner = None if other_component is None else _maybe_unpack_vector_component(plain(other_component))
        name = face_concat_name(grid, da, partner, {ax_name: (lo, hi)})
    for m in (m_in, m_out):  # xarray's name rule for the explicit `* metric` / `/ metric` of the reference
        if m is not None and getattr(m, "name", None) != name:
            name = None
    res = LazyArray(node, out_dims, shape, dt, _is_host(da), name=name)
    res = _reattach_coords([res], grid, ufunc.padding_width, {out_dim}, [da])[0]
    _pending_add(grid, node)
    _count("deferred_stencil")
    return res


# ==============================================================================================
# binary nodes: xarray-style `+ - * /` with a deferred operand
# ==============================================================================================
class BinaryNode(Node):
    __slots__ = ("op", "a", "b", "reflexive", "dims_order", "dims")

    def __init__(self, op, a, b, reflexive, dims_order, dims):
        self.value = None
        self.op, self.a, self.b, self.reflexive, self.dims_order, self.dims = op, a, b, reflexive, dims_order, tuple(dims)

    def _release(self) -> None:
        self.a = self.b = None

    def eager(self):
        a = plain(self.a)
        return DataArray._binary(a, plain(self.b), self.op, self.reflexive, self.dims_order).data

    def _compute(self):
        for rule in (_try_curl_or_div, _try_flux, _try_stencil_over_metric):
            got = rule(self)
            if got is not None:
                return got
        _count("binary_eager")
        return self.eager()


_NP_OP = {"mul": np.multiply, "div": np.divide, "add": np.add, "sub": np.subtract}

_ACTIVE = threading.local()  # depth of `with grid.fused():` blocks on this thread (any grid)


def enter() -> None:
    _ACTIVE.depth = getattr(_ACTIVE, "depth", 0) + 1


def leave() -> None:
    _ACTIVE.depth = getattr(_ACTIVE, "depth", 1) - 1


def _hook(self_operand, other, op, reflexive, dims_order):
    """DataArray._binary's way in: an ordinary array next to a deferred result stays deferred; inside a `grid.fused()`
    block `field * metric` between two ORDINARY arrays is deferred as well, so that a stencil applied to the product can
    take the metric into its own launch (`grid.diff(u * dy, "X")`).  None -> compute now."""
    if getattr(other, "is_deferred", False):
        return defer_binary(self_operand, other, op, reflexive, dims_order)
    if (op == "mul" and getattr(_ACTIVE, "depth", 0) > 0 and isinstance(other, DataArray)
            and not isinstance(self_operand, LazyArray) and not isinstance(other, LazyArray) and dims_order is None):
        a, m = (self_operand, other) if set(other.dims) <= set(self_operand.dims) else (other, self_operand)
        if (set(m.dims) <= set(a.dims) and tuple(self_operand.dims + tuple(d for d in other.dims if d not in self_operand.dims)) == tuple(a.dims)
                and _dtype_of(a) == _dtype_of(m) and _dtype_of(a) in _FUSABLE_FLOATS and _is_host(a) == _is_host(m)):
            return defer_binary(self_operand, other, op, reflexive, dims_order, force=True)
    return None


def defer_binary(self_operand, other, op: str, reflexive: bool, dims_order, force: bool = False) -> Optional[LazyArray]:
    """`self_operand OP other` (reflexive: `other OP self_operand`) as a LazyArray when one of them is a deferred result;
    None -> evaluate now.  Metadata (dims, coords, name, broadcast shape, dtype) is `DataArray._binary`'s."""
    lazy_ops = [x for x in (self_operand, other) if _unforced(x) is not None]
    if (not lazy_ops and not force) or op not in _NP_OP:
        return None
    a = self_operand
    if isinstance(other, (int, float, np.integer, np.floating)):
        dims, coords = a.dims, OrderedDict(a.coords)
        shape = a.shape
        rt = _NP_OP[op](np.ones(1, _dtype_of(a)), other).dtype
        host = _is_host(a)
    elif isinstance(other, DataArray):
        dims = a.dims + tuple(d for d in other.dims if d not in a.dims)
        if dims_order is not None:
            dims = tuple(d for d in dims_order if d in dims) + tuple(d for d in dims if d not in dims_order)
        sa, sb = a.sizes, other.sizes
        for d in dims:
            if d in sa and d in sb and sa[d] != sb[d]:
                raise ValueError(f"cannot broadcast: dimension {d!r} has sizes {sa[d]} and {sb[d]}")
        shape = tuple(sa[d] if d in sa else sb[d] for d in dims)
        coords = OrderedDict(a.coords)
        for k, v in other.coords.items():  # (the eager rule, labeled.DataArray._binary: conflicting non-index coordinates go)
            mine = coords.get(k)
            if mine is None:
                coords[k] = v
            elif k not in dims and not _labeled._same_coord(mine, v):
                del coords[k]
        rt = _NP_OP[op](np.ones(1, _dtype_of(a)), np.ones(1, _dtype_of(other))).dtype
        host = _is_host(a) and _is_host(other)
    else:
        return None
    node = BinaryNode(op, a, other, reflexive, dims_order, dims)
    for x in (a, other):  # sibling rules (flux) look pending products up through the grid of their stencil operand
        n = _unforced(x, StencilNode)
        if n is not None:
            _pending_add(n.grid, node)
    _count("deferred_binary")
    return LazyArray(node, dims, shape, rt, host, coords=coords, name=_labeled._result_name(a, other))


# ==============================================================================================
# the rules
# ==============================================================================================
def _uniform(*arrays) -> Optional[np.dtype]:
    """the one float32 / float64 dtype all of `arrays` share, else None"""
    dts = {_dtype_of(a) for a in arrays if a is not None}
    if len(dts) != 1:
        return None
    dt = dts.pop()
    return dt if dt in _FUSABLE_FLOATS else None


def _finish(out, host: bool):
    return _dev.tohost(out) if host else out


def _as_pair(node: BinaryNode):
    """(first operand, second operand) of `first OP second` as written"""
    return (node.b, node.a) if node.reflexive else (node.a, node.b)


def _try_curl_or_div(root: BinaryNode):
    """`(diff(v, X) - diff(u, Y)) [/ area]` -> xg_vorticity; `(diff(u, X) + diff(v, Y)) [/ area]` -> xg_divergence"""
    area = None
    inner = root
    if root.op == "div":
        first, second = _as_pair(root)
        inner = _unforced(first, BinaryNode)
        if inner is None or not isinstance(second, DataArray) or _unforced(second) is not None:
            return None
        area = plain(second)
        if tuple(root.dims) != tuple(inner.dims) or not set(area.dims) <= set(inner.dims):
            return None
    if inner.op not in ("sub", "add"):
        return None
    first, second = _as_pair(inner)
    s1, s2 = _unforced(first, StencilNode), _unforced(second, StencilNode)
    if s1 is None or s2 is None or s1.grid is not s2.grid or s1.ax_name == s2.ax_name:
        return None
    if tuple(first.dims) != tuple(second.dims) or tuple(first.dims) != tuple(inner.dims):
        return None
    if inner.op == "sub":
        if not (s1.plain_diff("center", "left") and s2.plain_diff("center", "left")):
            return None
        # (v along X) - (u along Y): the contiguous dim of the result is X
        sx, sy = s1, s2
        v, u = plain(sx.source), plain(sy.source)
        ok = (len(u.dims) >= 2 and u.dims[-2:] == (sy.in_dim, sx.out_dim) and v.dims[-2:] == (sy.out_dim, sx.in_dim)
              and u.dims[:-2] == v.dims[:-2])
        widths = (1, 0)
        kernel, rule = _dev.vorticity, "vorticity"
    else:
        if not (s1.plain_diff("left", "center") and s2.plain_diff("left", "center")):
            return None
        # one difference along the contiguous dim (X, of u), the other along the dim before it (Y, of v): either order
        sx, sy = (s1, s2) if s1.source.dims[-1:] == (s1.in_dim,) else (s2, s1)
        u, v = plain(sx.source), plain(sy.source)
        ok = (len(u.dims) >= 2 and u.dims[-2:] == (sy.out_dim, sx.in_dim) and v.dims[-2:] == (sy.in_dim, sx.out_dim)
              and u.dims[:-2] == v.dims[:-2])
        widths = (0, 1)
        kernel, rule = _dev.divergence, "divergence"
    if not ok or _uniform(u, v, area) is None or _is_host(u) != _is_host(v):
        return None
    out_dims = u.dims[:-2] + (sy.out_dim, sx.out_dim)
    if tuple(out_dims) != tuple(inner.dims):
        return None
    grid = sx.grid
    if area is not None:
        area = _aligned_view(grid._resident(area, u.data), out_dims)
    bcx, fx, hx = sx.halo(widths)
    bcy, fy, hy = sy.halo(widths)
    out = kernel(u.data, v.data, area, bcx, bcy, fx, fy, hx, hy)
    _count(rule)
    return _finish(out, _is_host(u))


def _try_stencil_over_metric(root: BinaryNode):
    """`op(a, X) / n` with the stencil unevaluated and `n` broadcast along its dims -> the same launch with n as m_out"""
    if root.op != "div":
        return None
    first, second = _as_pair(root)
    s = _unforced(first, StencilNode)
    if s is None or s.m_out is not None or not isinstance(second, DataArray) or _unforced(second) is not None:
        return None
    n = plain(second)
    if tuple(root.dims) != tuple(s.out_dims) or not set(n.dims) <= set(s.out_dims):
        return None
    if _uniform(s.source, n) is None:
        return None
    n = s.grid._resident(n, like(s.source))
    return s.run(extra_m_out=n)


def _siblings(node, cls):
    grid = node.grid if isinstance(node, StencilNode) else None
    if grid is None:
        return []
    return [n for n in _pending_list(grid) if n is not node and isinstance(n, cls) and n.value is None]


def _try_gradient(s: StencilNode):
    """`diff(a, X)` and `diff(a, Y)` (or the two `derivative`s) of ONE centre field, both still unevaluated: one launch of
    xg_gradient reads the field once and writes both"""
    if not (s.funcname == "diff" and s.m_in is None and (s.ufunc.from_pos, s.ufunc.to_pos) == ("center", "left")):
        return None
    if s.other_component is not None or isinstance(s.arg, dict):
        return None
    a = s.source
    if len(a.dims) < 2 or s.in_dim not in a.dims[-2:]:
        return None
    key = _source_key(a)
    for t in _siblings(s, StencilNode):
        if (t.funcname == "diff" and t.m_in is None and (t.ufunc.from_pos, t.ufunc.to_pos) == ("center", "left")
                and t.other_component is None and not isinstance(t.arg, dict) and t.ax_name != s.ax_name
                and _source_key(t.source) == key and t.in_dim in a.dims[-2:] and t.in_dim != s.in_dim
                and tuple(t.source.dims) == tuple(a.dims)):
            sx, sy = (s, t) if a.dims[-1] == s.in_dim else (t, s)
            field = plain(a)
            if _uniform(field, sx.m_out, sy.m_out) is None:
                continue
            dims_x = field.dims[:-1] + (sx.out_dim,)
            dims_y = field.dims[:-2] + (sy.out_dim, field.dims[-1])
            grid = s.grid
            mx = None if sx.m_out is None else _aligned_view(grid._resident(plain(sx.m_out), field.data), dims_x)
            my = None if sy.m_out is None else _aligned_view(grid._resident(plain(sy.m_out), field.data), dims_y)
            bcx, fx, hx = sx.halo((1, 0))
            bcy, fy, hy = sy.halo((1, 0))
            gx, gy = _dev.gradient(field.data, bcx, bcy, fx, fy, mx, my, hx, hy)
            host = _is_host(field)
            vx, vy = _finish(gx, host), _finish(gy, host)
            (sy if s is sx else sx).set(vy if s is sx else vx)   # the sibling has its value now; `s` gets its own from the caller
            _count("gradient")
            return vx if s is sx else vy
    return None


def _try_two_axes(s: StencilNode):
    """`op(op(a, A), B)` with the inner result unevaluated -> the two-axis kernel (Grid._two_axes_in_one_pass)"""
    inner = _unforced(s.arg, StencilNode)
    if inner is None or inner.funcname != s.funcname or inner.ax_name == s.ax_name:
        return None
    if any(m is not None for m in (s.m_in, s.m_out, inner.m_in, inner.m_out)):
        return None
    if s.other_component is not None or inner.other_component is not None or isinstance(inner.arg, dict):
        return None
    if inner.remaining != s.remaining or s.complex or inner.complex:
        return None
    a = plain(inner.source)
    res = s.grid._two_axes_in_one_pass(s.funcname, a, (inner.sig, inner.ax_name), (s.sig, s.ax_name), dict(s.remaining))
    if res is None or tuple(res.dims) != tuple(s.out_dims):
        return None
    _count("two_axes")
    return res.data


def _try_flux(root: BinaryNode):
    """`u * interp(T, X)` and `v * interp(T, Y)` of ONE centre tracer, both unevaluated: xg_flux reads T once"""
    mine = _flux_parts(root)
    if mine is None:
        return None
    comp, s = mine
    t = s.source
    key = _source_key(t)
    for other in [n for n in _pending_list(s.grid) if n is not root and isinstance(n, BinaryNode) and n.value is None]:
        parts = _flux_parts(other)
        if parts is None:
            continue
        comp2, s2 = parts
        if s2.ax_name == s.ax_name or _source_key(s2.source) != key or s2.grid is not s.grid:
            continue
        if tuple(t.dims[-2:]) not in ((s.in_dim, s2.in_dim), (s2.in_dim, s.in_dim)):
            continue
        (cx, sx, nx), (cy, sy, ny) = ((comp, s, root), (comp2, s2, other)) if t.dims[-1] == s.in_dim else \
            ((comp2, s2, other), (comp, s, root))
        tracer, u, v = plain(t), plain(cx), plain(cy)
        dims_x = tracer.dims[:-1] + (sx.out_dim,)
        dims_y = tracer.dims[:-2] + (sy.out_dim, tracer.dims[-1])
        if (tuple(u.dims) != dims_x or tuple(v.dims) != dims_y or tuple(nx.dims) != dims_x or tuple(ny.dims) != dims_y
                or _uniform(tracer, u, v) is None or len({_is_host(tracer), _is_host(u), _is_host(v)}) != 1):
            continue
        bcx, fx, hx = sx.halo((1, 0))
        bcy, fy, hy = sy.halo((1, 0))
        qx, qy = _dev.flux(u.data, v.data, tracer.data, bcx, bcy, fx, fy, hx, hy)
        host = _is_host(tracer)
        vx, vy = _finish(qx, host), _finish(qy, host)
        (ny if root is nx else nx).set(vy if root is nx else vx)
        _count("flux")
        return vx if root is nx else vy
    return None


def _flux_parts(node: BinaryNode):
    """(component, interp node) of `component * interp(T, axis)` (either order), else None"""
    if node.op != "mul":
        return None
    for comp, lazy in ((node.a, node.b), (node.b, node.a)):
        s = _unforced(lazy, StencilNode)
        if s is None or not isinstance(comp, DataArray) or _unforced(comp) is not None:
            continue
        if (s.funcname == "interp" and s.m_in is None and s.m_out is None and s.other_component is None
                and not isinstance(s.arg, dict) and (s.ufunc.from_pos, s.ufunc.to_pos) == ("center", "left")
                and len(s.source.dims) >= 2):
            return comp, s
    return None


from . import labeled as _labeled  # noqa: E402

_labeled._LAZY_HOOK = _hook
