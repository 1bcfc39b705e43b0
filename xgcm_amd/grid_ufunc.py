"""Grid-ufunc runtime: signatures, the `GridUFunc` plugin object, and `apply_as_grid_ufunc`.

This is the reference's operator/plugin interface for the hot path (xgcm/grid_ufunc.py):
a grid ufunc is a callable on UNLABELLED arrays whose core dims are last and already padded by
`padding_width`; it is registered with `@as_grid_ufunc(signature=..., padding_width=...)` and
called as `ufunc(grid, *args, axis=[(ax,)], **kw)`.  The generic path below keeps that contract
for user functions, with the padding done by the HIP pad kernel (`xgcm_amd.padding.pad`) instead
of `DataArray.pad`.  The built-in 1-D operators do not go through it: `xgcm_amd.gridops`
overrides `__call__` with a single fused kernel launch (halo + stencil + metrics).

dask is absent from the target image: `dask=` / `map_overlap=` are accepted for signature
compatibility and ignored (arrays here are never chunked; reference grid_ufunc.py:1045-1223 is
out of scope).
"""

from __future__ import annotations

import re
from collections import OrderedDict
from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Set, Tuple, Union

import numpy as np

from .labeled import DataArray, _is_tensor, from_xarray, is_xarray, to_xarray
from .padding import pad

_PAIR = re.compile(r"(\w+):(center|left|right|inner|outer)")
T_AX_POS_LIST = List[Tuple[str, ...]]


# ------------------------------------------------------------------------------------------
# signatures  (reference grid_ufunc.py:147-301; grammar in SURVEY.md A.7)
# ------------------------------------------------------------------------------------------
def _split_arguments(txt: str) -> Optional[List[List[Tuple[str, str]]]]:
    """'(a:left,b:center),(c:inner)' -> [[('a','left'),('b','center')],[('c','inner')]]; None if malformed."""
    if not txt:
        return None
    args: List[List[Tuple[str, str]]] = []
    i, n = 0, len(txt)
    while True:
        if i >= n or txt[i] != "(":
            return None
        j = txt.find(")", i)
        if j < 0:
            return None
        body = txt[i + 1:j]
        if "(" in body:
            return None
        pairs: List[Tuple[str, str]] = []
        if body:
            items = body.split(",")
            if items[-1] == "":
                items = items[:-1]  # a single trailing comma is tolerated
            for it in items:
                m = _PAIR.fullmatch(it)
                if m is None:
                    return None
                pairs.append((m.group(1), m.group(2)))
        args.append(pairs)
        i = j + 1
        if i == n:
            return args
        if txt[i] != ",":
            return None
        i += 1


def _parse_signature_from_string(signature: str):
    sig = signature.replace(" ", "")
    halves = sig.split("->")
    parsed = [_split_arguments(h) for h in halves] if len(halves) == 2 else [None]
    if any(p is None for p in parsed):
        raise ValueError(f"Not a valid grid ufunc signature: {sig}")
    (ins, outs) = parsed
    names = lambda side: [tuple(n for n, _ in arg) for arg in side]  # noqa: E731
    poss = lambda side: [tuple(p for _, p in arg) for arg in side]  # noqa: E731
    return names(ins), poss(ins), names(outs), poss(outs)


class _GridUFuncSignature:
    """Axis names (dummy variables) and grid positions of every input and output of a grid ufunc."""

    def __init__(self, in_ax_names: T_AX_POS_LIST, in_ax_positions: T_AX_POS_LIST, out_ax_names: T_AX_POS_LIST,
                 out_ax_positions: T_AX_POS_LIST):
        if not in_ax_names or not in_ax_positions:
            raise ValueError(
                "At least one input argument of the Grid UFunc signature must have axis names and positions"
            )
        self.in_ax_names = in_ax_names
        self.in_ax_positions = in_ax_positions
        self.out_ax_names = out_ax_names
        self.out_ax_positions = out_ax_positions

    @classmethod
    def from_string(cls, signature: str) -> "_GridUFuncSignature":
        return cls(*_parse_signature_from_string(signature))

    @classmethod
    def from_type_hints(cls, hints: Dict[str, Any]) -> "_GridUFuncSignature":
        """From `typing.get_type_hints(ufunc, include_extras=True)`: every argument / return value
        annotated as `Annotated[np.ndarray, "X:center,Y:left"]` contributes one signature argument
        (reference grid_ufunc.py:303-361)."""
        hints = dict(hints)
        if "return" in hints:
            out_anns = [_annotation_of(h) for h in _return_hints(hints.pop("return"))]
            out_anns = [a for a in out_anns if a is not None]
            out_names = [tuple(n for n, _ in _PAIR.findall(a)) for a in out_anns]
            out_pos = [tuple(p for _, p in _PAIR.findall(a)) for a in out_anns]
        else:
            out_names, out_pos = [()], [()]
        in_anns = [a for a in (_annotation_of(h) for h in hints.values()) if a is not None]
        in_names = [tuple(n for n, _ in _PAIR.findall(a)) for a in in_anns]
        in_pos = [tuple(p for _, p in _PAIR.findall(a)) for a in in_anns]
        sig = cls(in_names, in_pos, out_names, out_pos)
        _parse_signature_from_string(str(sig))  # sanity check, raises "Not a valid grid ufunc signature"
        return sig

    def __str__(self) -> str:
        def side(names, poss):
            return ",".join("(" + ",".join(f"{n}:{p}" for n, p in zip(ns, ps)) + ")" for ns, ps in zip(names, poss))

        return f"{side(self.in_ax_names, self.in_ax_positions)}->{side(self.out_ax_names, self.out_ax_positions)}"

    __repr__ = __str__

    def _canonical(self):
        """Structure with dummy axis names replaced by their order of first appearance."""
        order: Dict[str, int] = {}
        for arg in list(self.in_ax_names) + list(self.out_ax_names):
            for n in arg:
                order.setdefault(n, len(order))
        canon = lambda side: tuple(tuple(order[n] for n in arg) for arg in side)  # noqa: E731
        return (canon(self.in_ax_names), tuple(map(tuple, self.in_ax_positions)), canon(self.out_ax_names),
                tuple(map(tuple, self.out_ax_positions)))

    def equivalent(self, other: "_GridUFuncSignature") -> bool:
        """Equal up to a consistent renaming of the dummy axis names; positions must match exactly."""
        return self._canonical() == other._canonical()


def _annotation_of(hint) -> Optional[str]:
    meta = getattr(hint, "__metadata__", None)
    return meta[0] if meta else None


def _return_hints(hint) -> list:
    """`Tuple[Annotated[...], Annotated[...]]` -> its members; anything else -> [hint]."""
    import typing

    if typing.get_origin(hint) is tuple:
        return list(typing.get_args(hint))
    return [hint]


def _signature_from_str_or_type_hints(ufunc, str_sig) -> "_GridUFuncSignature":
    """Axis positions come from the `signature` kwarg or from `Annotated` type hints, never both
    (reference grid_ufunc.py:472-502)."""
    import typing

    if isinstance(str_sig, _GridUFuncSignature):
        return str_sig
    try:
        hints = typing.get_type_hints(ufunc, include_extras=True)
    except Exception:  # builtins, partials, objects without introspectable annotations
        hints = {}
    annotated = any(_annotation_of(h) is not None for k, h in hints.items() if k != "return") or (
        "return" in hints and any(_annotation_of(h) is not None for h in _return_hints(hints["return"])))
    if str_sig:
        if annotated:
            raise ValueError(
                "Must specify axis positions through only one of either type hints or signature kwarg, not both."
            )
        return _GridUFuncSignature.from_string(str_sig)
    if not annotated:
        raise ValueError("Must specify axis positions through either type hints or signature kwarg")
    return _GridUFuncSignature.from_type_hints(hints)


# ------------------------------------------------------------------------------------------
# small helpers shared with grid.py
# ------------------------------------------------------------------------------------------
def _maybe_unpack_vector_component(data):
    if isinstance(data, dict):
        [da] = list(data.values())
        return da
    return data


def _check_data_input(data, grid):
    """Validate one data argument: a DataArray, or a one-entry {axis: DataArray} vector component."""
    if data is None:
        return data
    if not isinstance(data, (DataArray, dict)):
        raise TypeError(f"All data arguments must be either a DataArray or Dictionary Got {type(data)}.")
    if isinstance(data, dict):
        if len(data) != 1:
            raise ValueError(
                "Vector components provided as dictionaries should contain exactly one key/value pair."
                f" Found {len(data)}. Full input:{data}"
            )
        [(key, value)] = data.items()
        if key not in grid.axes:
            raise ValueError(
                f"Vector component with unknown axis provided. Grid has axes ({list(grid.axes)}), got  ({key})"
            )
        if not isinstance(value, DataArray):
            raise TypeError(f"Dictionary inputs must have a DataArray as value. Got {type(value)}.")
    return data


def _promote_to_sequence_and_check(data, grid):
    if not isinstance(data, Sequence) or isinstance(data, (str, bytes)):
        data = [data]
    return [_check_data_input(d, grid) for d in data]


def _identify_dummy_axes_with_real_axes(sig_in_dummy_ax_names, axis) -> Mapping[str, str]:
    if len(axis) != len(sig_in_dummy_ax_names):
        raise ValueError("Number of entries in `axis` does not match the number of variables in the input signature")
    for i, (real, dummy) in enumerate(zip(axis, sig_in_dummy_ax_names)):
        if len(real) != len(dummy):
            raise ValueError(
                f"Number of Axes in `axis` entry number {i} does not match the number of Axes in that entry in the input signature"
            )
    dummies = list(OrderedDict.fromkeys(ax for arg in sig_in_dummy_ax_names for ax in arg))
    reals = list(OrderedDict.fromkeys(ax for arg in axis for ax in arg))
    if len(dummies) != len(reals):
        raise ValueError(
            f"Found {len(dummies)} unique input axes in signature but {len(reals)} "
            f"real unique input axes were supplied to the grid ufunc when called"
        )
    return dict(zip(dummies, reals))


def _substitute_dummy_axis_names(padding_width, mapping):
    if padding_width:
        return {mapping[ax]: tuple(width) for ax, width in padding_width.items()}
    return {real: (0, 0) for real in mapping.values()}


def _reattach_coords(results, grid, padding_width, out_core_dim_names: Optional[Set[str]] = None, input_args=None):
    """Coords of grid._own_ds whose dims all survive, overridden by input coords on non-core dims
    (first input wins).  Reference grid_ufunc.py:1262-1320."""
    out_core_dim_names = out_core_dim_names or set()
    from_inputs: Dict[str, DataArray] = OrderedDict()
    for arg in input_args or []:
        for cname, c in arg.coords.items():
            if any(d in out_core_dim_names for d in c.dims):
                continue
            from_inputs.setdefault(cname, c)
    out = []
    for res in results:
        rdims = set(res.dims)
        chosen = OrderedDict((k, c) for k, c in grid._own_ds.coords.items() if all(d in rdims for d in c.dims))
        for k, c in from_inputs.items():
            if all(d in rdims for d in c.dims):
                chosen[k] = c
        try:
            res = res.assign_coords(chosen)
        except ValueError as err:
            if padding_width and str(err).startswith("conflicting sizes"):
                raise ValueError(
                    f"{str(err)} - does your grid ufunc correctly trim off the same number of elements "
                    f"which were added by padding using padding_width={padding_width}?"
                )
            raise
        out.append(res)
    return out


def _restore_input_dim_order(results, args, sig, in_core_dims, out_core_dims):
    """Outputs follow the inputs' dim order with shifted core dims renamed in place (GH #533)."""
    in_of = {ax: dim for names, dims in zip(sig.in_ax_names, in_core_dims) for ax, dim in zip(names, dims)}
    out_of = {ax: dim for names, dims in zip(sig.out_ax_names, out_core_dims) for ax, dim in zip(names, dims)}
    renamed = {in_of[ax]: out_of[ax] for ax in in_of if ax in out_of}
    reference_order: List[str] = []
    for arg in args:
        for d in _maybe_unpack_vector_component(arg).dims:
            d = renamed.get(d, d)
            if d not in reference_order:
                reference_order.append(d)
    fixed = []
    for res in results:
        order = [d for d in reference_order if d in res.dims] + [d for d in res.dims if d not in reference_order]
        fixed.append(res.transpose(*order))
    return tuple(fixed)


# ------------------------------------------------------------------------------------------
# GridUFunc / decorator
# ------------------------------------------------------------------------------------------
class GridUFunc:
    """A function on unlabelled arrays bound to a grid signature (reference grid_ufunc.py:373-562)."""

    def __init__(self, ufunc: Callable, **kwargs):
        self.ufunc = ufunc
        if "boundary" in kwargs:
            raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
        if "boundary_width" in kwargs:
            raise ValueError(
                "Argument 'boundary_width' has been renamed to 'padding_width'. Please use 'padding_width' instead."
            )
        self.signature = _signature_from_str_or_type_hints(ufunc, kwargs.pop("signature", ""))
        self.padding_width = kwargs.pop("padding_width", None)
        self.padding = kwargs.pop("padding", None)
        self.fill_value = kwargs.pop("fill_value", None)
        self.dask = kwargs.pop("dask", "forbidden")
        self.map_overlap = kwargs.pop("map_overlap", False)
        self.pad_before_func = kwargs.pop("pad_before_func", True)
        if kwargs:
            raise TypeError(f"Unsupported keyword argument(s) provided: {list(kwargs.keys())}")

    @property
    def boundary(self):
        raise AttributeError("Attribute 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")

    @property
    def boundary_width(self):
        raise AttributeError(
            "Attribute 'boundary_width' has been renamed to 'padding_width'. Please use 'padding_width' instead."
        )

    def __repr__(self) -> str:
        return f"GridUFunc(ufunc={self.ufunc}, signature='{self.signature}', padding_width='{self.padding_width}')"

    def __call__(self, grid=None, *args, axis, **kwargs):
        if "boundary" in kwargs:
            raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
        # call-time kwargs beat the values bound by the decorator (grid_ufunc.py:544-548)
        return apply_as_grid_ufunc(
            self.ufunc,
            *args,
            axis=axis,
            grid=grid,
            signature=self.signature,
            padding_width=self.padding_width,
            padding=kwargs.pop("padding", self.padding),
            fill_value=kwargs.pop("fill_value", self.fill_value),
            dask=kwargs.pop("dask", self.dask),
            map_overlap=kwargs.pop("map_overlap", self.map_overlap),
            pad_before_func=kwargs.pop("pad_before_func", self.pad_before_func),
            **kwargs,
        )


_ALLOWED_DECORATOR_KWARGS = {"padding", "fill_value", "dask", "map_overlap", "pad_before_func"}


def as_grid_ufunc(signature: str = "", padding_width=None, **kwargs) -> Callable:
    """Decorator turning a function on unlabelled arrays into a grid-aware ufunc."""
    if "boundary" in kwargs:
        raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
    if "boundary_width" in kwargs:
        raise ValueError(
            "Argument 'boundary_width' has been renamed to 'padding_width'. Please use 'padding_width' instead."
        )
    bad = list(kwargs.keys() - _ALLOWED_DECORATOR_KWARGS)
    if bad:
        raise TypeError(f"Unsupported keyword argument(s) provided: {bad}")

    def bind(ufunc):
        return GridUFunc(ufunc, signature=signature, padding_width=padding_width, **kwargs)

    return bind


# ------------------------------------------------------------------------------------------
# generic application of a user function
# ------------------------------------------------------------------------------------------
def _move_core_last(da: DataArray, core_dims: Sequence[str], bdims: Sequence[str]):
    """Unlabelled view of `da` ordered (broadcast dims..., core dims...), size-1 where a broadcast dim is absent."""
    present = [d for d in bdims if d in da.dims]
    v = da.transpose(*present, *core_dims).data
    index = tuple(slice(None) if d in da.dims else None for d in bdims) + (slice(None),) * len(core_dims)
    return v[index]


def _apply(func, args: Sequence[DataArray], in_core_dims, out_core_dims, **kwargs):
    """What `xr.apply_ufunc(func, *args, input_core_dims, output_core_dims, exclude_dims)` does for
    in-memory arrays: broadcast the non-core dims by name, core dims last, call, relabel."""
    core = set(d for a in in_core_dims for d in a) | set(d for a in out_core_dims for d in a)
    bdims: List[str] = []
    bsize: Dict[str, int] = {}
    for a in args:
        for d, s in a.sizes.items():
            if d in core:
                continue
            if d not in bdims:
                bdims.append(d)
                bsize[d] = s
            elif bsize[d] != s:
                raise ValueError(f"operands could not be broadcast together on dimension {d!r}")
    raw = [_move_core_last(a, cd, bdims) for a, cd in zip(args, in_core_dims)]
    names = {a.name for a in args}
    name = names.pop() if len(names) == 1 else None  # xarray: an output keeps the name all the inputs share
    result = func(*raw, **kwargs)
    if not isinstance(result, tuple):
        result = (result,)
    if len(result) != len(out_core_dims):
        raise ValueError(
            f"applied function returned {len(result)} outputs but the signature declares {len(out_core_dims)}"
        )
    out = []
    for r, ocd in zip(result, out_core_dims):
        if not _is_tensor(r):
            r = np.asarray(r)
        dims = tuple(bdims) + tuple(ocd)
        if len(r.shape) != len(dims):
            raise ValueError(
                f"applied function returned data with {len(r.shape)} dimensions, expected {len(dims)}: {dims}"
            )
        out.append(DataArray(r, dims, name=name))
    return tuple(out)


def apply_as_grid_ufunc(func: Callable, *args, axis=None, grid=None, signature: Union[str, _GridUFuncSignature] = "",
                        padding_width=None, padding=None, fill_value=None, dask: str = "forbidden",
                        map_overlap: bool = False, pad_before_func: bool = True, other_component=None, **kwargs):
    """Apply `func` to the arguments in a grid-aware manner (reference grid_ufunc.py:661-951)."""
    if "boundary" in kwargs:
        raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
    if "boundary_width" in kwargs:
        raise ValueError(
            "Argument 'boundary_width' has been renamed to 'padding_width'. Please use 'padding_width' instead."
        )
    if "keep_coords" in kwargs:
        raise ValueError(
            "The 'keep_coords' argument has been removed. Coordinates compatible with the output are now always preserved."
        )
    if grid is None:
        raise ValueError("Must provide a grid object to describe the Axes")

    # xarray objects in -> xarray objects out (the reference's surface); everything in between is labelled
    # arrays of this package
    def _unwrap(a):
        if isinstance(a, dict):
            return {k: _unwrap(v) for k, v in a.items()}
        return from_xarray(a) if is_xarray(a) else a

    def _any_xr(seq):
        return any(is_xarray(v) for a in seq for v in (a.values() if isinstance(a, dict) else (a,)))

    others = other_component if isinstance(other_component, (list, tuple)) else [other_component]
    was_xr = _any_xr(args) or _any_xr(o for o in others if o is not None)
    if was_xr:
        res = apply_as_grid_ufunc(func, *[_unwrap(a) for a in args], axis=axis, grid=grid, signature=signature,
                                  padding_width=padding_width, padding=padding, fill_value=fill_value, dask=dask,
                                  map_overlap=map_overlap, pad_before_func=pad_before_func,
                                  other_component=([_unwrap(o) for o in other_component] if isinstance(other_component, (list, tuple))
                                                   else _unwrap(other_component)) if other_component is not None else None,
                                  **kwargs)
        return tuple(to_xarray(r) for r in res) if isinstance(res, tuple) else to_xarray(res)

    args = _promote_to_sequence_and_check(list(args), grid)
    other_component = _promote_to_sequence_and_check(other_component, grid)
    if len(other_component) == 1 and other_component[0] is None:
        other_component = other_component * len(args)
    if len(args) != len(other_component):
        raise ValueError(
            "When providing multiple input arguments, `other_component` needs to provide one dictionary per input."
        )
    if axis is None:
        raise ValueError("Must provide an axis along which to apply the grid ufunc")
    if len(args) != len(axis):
        raise ValueError("Number of entries in `axis` does not match the number of data arguments supplied")

    sig = signature if isinstance(signature, _GridUFuncSignature) else _GridUFuncSignature.from_string(signature)
    real_of = _identify_dummy_axes_with_real_axes(sig.in_ax_names, axis)
    out_ax_names = [[real_of[ax] for ax in arg] for arg in sig.out_ax_names]

    unpacked = [_maybe_unpack_vector_component(a) for a in args]
    for i, (names, positions, arg) in enumerate(zip(axis, sig.in_ax_positions, unpacked)):
        for n, p in zip(names, positions):
            try:
                dim = grid.axes[n].coords[p]
            except KeyError:
                raise ValueError(f"Axis position ({n}:{p}) does not exist in grid")
            if dim not in arg.dims:
                raise ValueError(
                    f"Mismatch between signature and input argument {i}: "
                    f"Signature specified data to lie at Axis Position ({n}:{p}), "
                    f"but the corresponding grid coordinate {dim} "
                    f"does not appear in argument"
                    f"{arg}"
                )

    in_core_dims = [[grid.axes[n].coords[p] for n, p in zip(ns, ps)] for ns, ps in zip(axis, sig.in_ax_positions)]
    out_core_dims = [[grid.axes[n].coords[p] for n, p in zip(ns, ps)] for ns, ps in zip(out_ax_names, sig.out_ax_positions)]
    widths = _substitute_dummy_axis_names(padding_width, real_of)

    def pad_all(arrays, ocs):
        return [pad(a, grid=grid, padding_width=widths, padding=padding, fill_value=fill_value, other_component=oc)
                for a, oc in zip(arrays, ocs)]

    if pad_before_func:
        padded = [_maybe_unpack_vector_component(p) for p in pad_all(args, other_component)]
        results = _apply(func, padded, in_core_dims, out_core_dims, **kwargs)
    else:
        unpadded = _apply(func, unpacked, in_core_dims, out_core_dims, **kwargs)
        # the reference pairs the unpadded RESULTS with the per-INPUT `other_component` list (grid_ufunc.py:915-923 ->
        # :1016-1026, a `zip`): a function with more outputs than inputs loses the surplus outputs there, and so here
        results = tuple(pad_all(unpadded, other_component))

    out_core_dim_names = set(d for arg in out_core_dims for d in arg)
    results = _reattach_coords(results, grid, padding_width, out_core_dim_names, unpacked)
    results = _restore_input_dim_order(results, args, sig, in_core_dims, out_core_dims)
    if len(results) == 1:
        (results,) = results
    return results
