"""The 41 built-in grid ufuncs of the reference's `xgcm.gridops`, backed by HIP kernels.

Same names, signatures, `padding_width`, `fill_value` and `pad_before_func` as reference
xgcm/gridops.py:27-278 (pinned by tests/golden/gridops_table.json), so `_select_grid_ufunc`
(name prefix + signature) finds them exactly as it finds the reference's.  Two levels:

* `.ufunc` -- the raw body on an already padded, unlabelled array with the core dim last
  (API-compatible plugin level; runs `xg_stencil1d_f64`/`xg_cumsum1d_f64` without a halo);
* `__call__(grid, da, axis=[(ax,)], padding=..., fill_value=...)` -- the FUSED level used by
  `Grid.diff/interp/min/max`: boundary halo, stencil and the optional metric multiply/divide
  are one kernel launch on the array in its original dim order (no pad copy, no transpose).
"""

from __future__ import annotations


import numpy as np

from . import device as _dev
from . import dtypes as _dt
from .grid_ufunc import (
    GridUFunc,
    _check_data_input,
    _maybe_unpack_vector_component,
    _reattach_coords,
)
from .labeled import DataArray, _aligned_view, _is_tensor
from .padding import halo_cells, no_boundary_error

# (from, to) -> padding_width of the two-point stencils (reference gridops.py:27-65)
_STENCIL_WIDTHS = {
    ("center", "left"): (1, 0),
    ("left", "center"): (0, 1),
    ("center", "right"): (0, 1),
    ("right", "center"): (1, 0),
    ("center", "outer"): (1, 1),
    ("outer", "center"): (0, 0),
    ("center", "inner"): (0, 0),
    ("inner", "center"): (1, 1),
}

# (from, to) -> (padding_width, pad_before_func, fill_value, drop_last)  (reference gridops.py:221-278)
_CUMSUM_TABLE = {
    ("center", "left"): ((1, 0), False, 0, True),
    ("left", "center"): ((0, 0), True, None, False),
    ("center", "right"): ((0, 0), True, None, False),
    ("right", "center"): ((1, 0), False, 0, True),
    ("center", "outer"): ((1, 0), False, 0, False),
    ("outer", "center"): ((0, 0), True, None, True),
    ("center", "inner"): ((0, 0), True, None, True),
    ("inner", "center"): ((1, 0), False, 0, False),
}


def _same_residency(data, out):
    """Host data in -> host data out (one PCIe round trip); HBM data stays in HBM."""
    return out if _is_tensor(data) else _dev.tohost(out)


def is_integer_data(data) -> bool:
    """integer or bool data (numpy or torch): numpy keeps such arrays integral through diff / min / max / cumsum / pad,
    and so does the device layer (int64 lanes, xgcm_amd.dtypes) -- callers only need this to keep integers off the
    float-only fused kernels"""
    return getattr(data, "dtype", None) is not None and _dt.is_integer(_dt.np_dtype(data))


def _stencil(data, op, *args):
    return _same_residency(data, _dev.stencil1d(op, data, *args))


def _cumsum(data, *args):
    return _same_residency(data, _dev.cumsum1d(data, *args))


def complex_topology(grid, ax_name: str) -> bool:
    """True when halos along `ax_name` do not follow from the array itself: an axis that takes part
    in a face connection (as the linked edge or as the neighbour's axis), or the fold axis of a north
    fold.  Other axes of such grids (the vertical of an LLC grid, the seam axis of a fold) keep the
    ordinary wrap / clamp / constant rules -- the reference's `_pad_face_connections` reduces to
    `_pad_basic` for them -- and therefore the fully fused kernels."""
    if getattr(grid, "_face_connections", None) is not None and ax_name in getattr(grid, "_connected_axes", ()):
        return True
    return ax_name in (getattr(grid, "_folds", None) or {})


class HipGridUFunc(GridUFunc):
    """A built-in 1-D grid ufunc whose labelled call is a single fused kernel launch."""

    def __init__(self, funcname: str, from_pos: str, to_pos: str, ufunc, **kwargs):
        super().__init__(ufunc, **kwargs)
        self.funcname = funcname
        self.from_pos = from_pos
        self.to_pos = to_pos
        self.__name__ = f"{funcname}_{from_pos}_to_{to_pos}"

    # -- fused path ------------------------------------------------------------------------
    def _fusable(self, grid, args, axis, kwargs) -> bool:
        if grid is None or len(args) != 1 or axis is None or len(axis) != 1 or len(axis[0]) != 1:
            return False
        extra = set(kwargs) - {"padding", "fill_value", "dask", "map_overlap", "other_component", "pad_before_func",
                               "metric_in", "metric_out"}
        if extra:
            return False
        if kwargs.get("pad_before_func", self.pad_before_func) != self.pad_before_func:
            return False
        if complex_topology(grid, axis[0][0]):
            # halos come from neighbouring faces / the folded row: gathered first, then the same kernels
            # (xg_stencil1d_halo); scans keep the reference's scan-then-pad (Grid.cumsum has its own one-pass route).  An
            # input metric rides along for scalar fields (the product's halo = the two halos' product, xg_stencil1d_halo_w);
            # vector components with a metric keep the explicit product (their halos mix components)
            if self.funcname == "cumsum":
                return False
            if kwargs.get("metric_in") is not None:
                return not isinstance(args[0], dict) and kwargs.get("other_component") is None
        return True

    def __call__(self, grid=None, *args, axis, **kwargs):
        if not self._fusable(grid, args, axis, kwargs):
            if kwargs.get("metric_in") is not None or kwargs.get("metric_out") is not None:
                raise NotImplementedError("metric-fused call needs the single-array 1-D form")
            kwargs.pop("metric_in", None)
            kwargs.pop("metric_out", None)
            return super().__call__(grid, *args, axis=axis, **kwargs)
        return self._fused(grid, args[0], axis[0][0], **kwargs)

    def _fused(self, grid, arg, ax_name: str, padding="__default__", fill_value="__default__", metric_in=None,
               metric_out=None, other_component=None, **_ignored):
        arg = _check_data_input(arg, grid)
        da = _maybe_unpack_vector_component(arg)
        if padding == "__default__":
            padding = self.padding
        if fill_value == "__default__":
            fill_value = self.fill_value
        try:
            in_dim = grid.axes[ax_name].coords[self.from_pos]
        except KeyError:
            raise ValueError(f"Axis position ({ax_name}:{self.from_pos}) does not exist in grid")
        if in_dim not in da.dims:
            raise ValueError(
                f"Mismatch between signature and input argument 0: "
                f"Signature specified data to lie at Axis Position ({ax_name}:{self.from_pos}), "
                f"but the corresponding grid coordinate {in_dim} "
                f"does not appear in argument"
                f"{da}"
            )
        # (a target position the axis does not have is the reference's bare KeyError(position): only the INPUT positions
        # are checked with a message, xgcm/grid_ufunc.py:827-832)
        out_dim = grid.axes[ax_name].coords[self.to_pos]

        (lo, hi) = next(iter(self.padding_width.values())) if self.padding_width else (0, 0)
        num = da.get_axis_num(in_dim)
        out_dims = tuple(out_dim if d == in_dim else d for d in da.dims)
        m_in = None if metric_in is None else _aligned_view(metric_in, da.dims)
        m_out = None if metric_out is None else _aligned_view(metric_out, out_dims)
        if complex_topology(grid, ax_name) and (lo or hi):
            # the halo cells alone (a (lo+hi)-wide slab) are gathered through the token map, then
            # the ordinary kernel reads the field once: no padded copy (reference: pad, then apply)
            halo = halo_cells(arg, grid, ax_name, (lo, hi), padding=padding, fill_value=fill_value,
                              other_component=other_component)
            slab_name = halo.name  # (`halo_cells` names its slab as the reference's pad would name the array: face_concat_name)
            if metric_in is not None:
                # the reference multiplies, then pads the PRODUCT through the topology (grid.py:804-808): its halo cells are
                # field[src] * metric[src] -- the product of the two halo slabs (a fill cell stays the fill value: the
                # metric's slab holds 1 there) -- so the field is read once, weighted inside the kernel
                mh = metric_in
                if in_dim not in mh.dims:
                    # a metric without the operator's dim still changes from face to face (and along the other axis of an
                    # axis-swapping link): give it that dim (a small broadcast plane) so that it travels through the topology
                    zeros = np.zeros(da.sizes[in_dim], dtype=_dt.float_of(_dt.np_dtype(mh.data)))
                    mh = mh + DataArray(zeros, (in_dim,))
                mh = halo_cells(mh, grid, ax_name, (lo, hi), padding=padding, fill_value=1.0)
                halo = halo._binary(mh, "mul", dims_order=da.dims)
            data = _same_residency(da.data, _dev.stencil1d_halo(self.funcname, da.data, halo.data, num, lo, hi, m_out, m_in))
            res = DataArray(data, out_dims, name=slab_name if metric_in is None else da.name)
            return _reattach_coords([res], grid, self.padding_width, {out_dim}, [da])[0]

        bc = grid._complete_user_kwargs_using_axis_defaults(padding, "padding")[ax_name]
        fv = grid._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")[ax_name]
        fv = 0.0 if fv is None else fv  # cast to the array's dtype by the device layer, like numpy.pad does
        if (lo or hi) and bc is None:
            raise no_boundary_error(ax_name)
        faces = getattr(grid, "_face_connections", None)
        if faces is not None and (lo or hi) and grid._facedim in da.dims:
            # an axis no link touches keeps the fused kernels, but the reference pads it through its face loop all the same
            # (xgcm/padding.py:849-857, :394-396): a face the connections leave out is its KeyError here too
            for i in range(da.sizes[grid._facedim]):
                if i not in faces[grid._facedim]:
                    raise KeyError(i)
        if not (lo or hi):
            bc = None
        if self.funcname == "cumsum":
            _, _, _, drop_last = _CUMSUM_TABLE[(self.from_pos, self.to_pos)]
            data = _cumsum(da.data, num, 0, 1 if drop_last else 0, lo, hi, bc, fv, False, False, m_in, m_out)
        else:
            data = _stencil(da.data, self.funcname, num, lo, hi, bc, fv, m_in, m_out)
        res = DataArray(data, out_dims, name=da.name)
        return _reattach_coords([res], grid, self.padding_width, {out_dim}, [da])[0]


def _raw_stencil(op: str):
    def body(a):
        """Raw two-point body on a padded array, core dim last (reference gridops.py:23-24,76-77,123-175)."""
        return _stencil(a, op, -1, 0, 0, None, 0.0)

    body.__name__ = f"{op}_forward"
    return body


# the raw bodies under the reference's names (gridops.py:23, :76, :123, :172): a padded array in, the two-point result out
diff_forward, interp_forward = _raw_stencil("diff"), _raw_stencil("interp")
pairwise_forward_min, pairwise_forward_max = _raw_stencil("min"), _raw_stencil("max")


def _raw_cumsum(drop_last: bool):
    def body(a):
        """np.cumsum(a, -1)[..., :-1 if drop_last] (reference gridops.py:227-278); NaN propagates."""
        return _cumsum(a, -1, 0, 1 if drop_last else 0, 0, 0, None, 0.0, False, False)

    body.__name__ = "cumsum_trimmed" if drop_last else "cumsum_full"
    return body


def _register():
    ns = globals()
    for funcname in ("diff", "interp", "min", "max"):
        raw = _raw_stencil(funcname)
        for (f, t), width in _STENCIL_WIDTHS.items():
            ns[f"{funcname}_{f}_to_{t}"] = HipGridUFunc(
                funcname, f, t, raw, signature=f"(X:{f})->(X:{t})", padding_width={"X": width}
            )
    for (f, t), (width, before, fill, drop_last) in _CUMSUM_TABLE.items():
        kw = {} if before else {"fill_value": fill, "pad_before_func": False}
        ns[f"cumsum_{f}_to_{t}"] = HipGridUFunc(
            "cumsum", f, t, _raw_cumsum(drop_last), signature=f"(X:{f})->(X:{t})", padding_width={"X": width}, **kw
        )

    def _diff_left_to_inner(a):  # registered but unimplemented in the reference too (gridops.py:68-70)
        raise NotImplementedError

    ns["diff_left_to_inner"] = GridUFunc(_diff_left_to_inner, signature="(X:left)->(X:inner)")


_register()
