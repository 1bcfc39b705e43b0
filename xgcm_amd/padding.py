"""Boundary padding of labelled arrays on the GPU (generic path for user grid ufuncs).

`pad` has the reference's signature and semantics for simple topologies (xgcm/padding.py:765-871
-> `_pad_basic` :575-616): per axis, in `padding_width` order, periodic = numpy 'wrap',
fill = 'constant', extend = 'edge'; all coordinates are stripped from the result; all-zero
widths return the input untouched; a padded axis without a boundary condition raises the
reference's ValueError.  The whole multi-axis pad is ONE kernel launch (xg_pad_f64), not one
array copy per axis.  Face connections and north folds are out of scope.

The built-in diff/interp/min/max/cumsum operators never call this: their halo is fused into
the stencil kernels (xgcm_amd/gridops.py).
"""

from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

from . import device as _dev
from .labeled import DataArray, _is_tensor

_XGCM_BOUNDARY_KWARG_TO_XARRAY_PAD_KWARG = {"periodic": "wrap", "fill": "constant", "extend": "edge"}


def no_boundary_error(ax: str) -> ValueError:
    """The reference's message for a padded axis with `padding=None` (padding.py:601-608)."""
    return ValueError(
        f"No boundary condition was specified for axis {ax!r}, but the "
        f"requested operation needs to pad it. Set a boundary condition, "
        f"e.g. ``padding='fill'`` (or 'extend'/'periodic'), on the Grid "
        f"(``Grid(..., padding=...)``) or pass ``padding=`` to the "
        f"grid method."
    )


def _strip_all_coords(obj):
    if isinstance(obj, dict):
        return {k: _strip_all_coords(v) for k, v in obj.items()}
    return obj._replace(coords={})


def pad(data, grid, padding_width: Optional[Dict[str, Tuple[int, int]]], padding=None, fill_value=None,
        other_component=None, **kwargs):
    """Pad `data` along the given grid axes according to the boundary conditions."""
    if "boundary" in kwargs:
        raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
    if "boundary_width" in kwargs:
        raise ValueError(
            "Argument 'boundary_width' has been renamed to 'padding_width'. Please use 'padding_width' instead."
        )
    padding = grid._complete_user_kwargs_using_axis_defaults(padding, "padding")
    fill_value = grid._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")

    if padding_width is None or all(tuple(w) == (0, 0) for w in padding_width.values()):
        return data
    if getattr(grid, "_face_connections", None) is not None:
        raise NotImplementedError("face connections are not supported by the MI355X backend (padding.py:260-572)")

    data = _strip_all_coords(data)
    if isinstance(data, dict):
        [data] = list(data.values())

    widths: Dict[int, Tuple[int, int]] = {}
    bc: Dict[int, Optional[str]] = {}
    fv: Dict[int, float] = {}
    for ax, w in padding_width.items():
        if all(x == 0 for x in w):
            continue
        _, dim = grid.axes[ax]._get_position_name(data)
        mode = padding[ax]
        if mode is None:
            raise no_boundary_error(ax)
        if isinstance(mode, Mapping):
            raise NotImplementedError("north-fold padding is not supported by the MI355X backend")
        if mode not in _XGCM_BOUNDARY_KWARG_TO_XARRAY_PAD_KWARG:
            raise KeyError(mode)
        num = data.get_axis_num(dim)
        widths[num] = (int(w[0]), int(w[1]))
        bc[num] = mode
        f = fill_value[ax]
        fv[num] = 0.0 if f is None else float(f)
    if not widths:
        return data
    host = not _is_tensor(data.data)
    out = _dev.pad_nd(data.data, widths, bc, fv)
    return DataArray(_dev.tohost(out) if host else out, data.dims, name=data.name)
