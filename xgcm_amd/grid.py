"""`Grid`: the public operator surface (diff / interp / min / max / cumsum / derivative / integrate /
cumint / average / get_metric ...) of the reference's `xgcm.Grid`, with every array pass executed
by one fused HIP kernel.

Reference call stacks and what replaces them here:

* `Grid.diff/interp/min/max` -> `_1d_grid_ufunc_dispatch` (xgcm/grid.py:728-836): per axis
  `array*metric` -> pad copy -> apply_ufunc -> `array/metric`  (5+ memory passes).  Here: per axis ONE
  launch of `xg_stencil1d_f64` with halo, metric multiply and metric divide fused.
* `Grid.derivative` (grid.py:1576-1578): diff then `/ dx`  ->  same launch with `m_out = dx`.
* `Grid.integrate` (grid.py:1598-1605): `(da*w).sum(dim)`  ->  `xg_reduce1d_f64` (first axis fused with
  the weight multiply).
* `Grid.cumsum` (grid.py:1183-1418): flip/cumsum/flip/trim/pad/`/metric`  ->  one `xg_cumsum1d_f64`.

Metadata logic (axis lookup, default shifts, per-axis kwargs, metric search, coordinate
re-attachment, error types and messages) is restated from the reference so that the parity tests
read like the reference's own.  Grids with face connections or a north fold take the generic
pad-then-apply route (halo gather `xg_gather_f64`, then the un-padded stencil kernel).  Out of scope
(dask-style chunked inputs are walked block by block: xgcm_amd.chunked).
"""

from __future__ import annotations

import contextlib
import functools
import inspect
import itertools
import operator
import threading
import warnings
from collections import OrderedDict
from typing import Any, Dict, Iterable, List, Mapping, Optional, Tuple

import numpy as np

from . import device as _dev
from . import dtypes as _dt
from . import gridops
from .axis import Axis
from .grid_ufunc import (
    GridUFunc,
    _check_data_input,
    _GridUFuncSignature,
    _maybe_unpack_vector_component,
    _reattach_coords,
    apply_as_grid_ufunc,
)
from . import lazy as _lazy
from .labeled import CHUNKED_INPUT_MESSAGE, DataArray, Dataset, _aligned_view, _is_tensor, from_xarray, is_xarray, to_xarray
from .metrics import iterate_axis_combinations
from .padding import FoldSpec, InteriorOf, halo_cells, no_boundary_error, pad


def _maybe_promote_str_to_list(a):
    return [a] if isinstance(a, str) else a


def _name_after(name, *metrics):
    """name of `array OP metric ...` by xarray's rule (kept only while every operand carries it): the fused kernels
    take the metrics into the operator's launch, the NAME still follows the reference's explicit products / quotients"""
    for m in metrics:
        if m is not None and getattr(m, "name", None) != name:
            return None
    return name


class _DimsOnly:
    """Stand-in carrying only `.dims`/`.name`: all `get_metric` needs of its `array` argument."""

    def __init__(self, dims, name=None):
        self.dims = tuple(dims)
        self.name = name


def _refuse_chunked_core_dim_with_changing_length(grid, da, ax_name, sig):
    """The reference maps a ufunc over the chunks of a chunked CORE dim with `dask.array.map_overlap` and refuses signatures
    whose positions change the dim's length (`xgcm/grid_ufunc.py:1136-1159`, reached from `xgcm/grid.py:810-816`; `cumsum` is
    exempt there).  Chunks along the operator's axis are read together here, so nothing would go wrong -- but the same call
    raises the same error, so that code written against this backend also runs on the reference."""
    chunks = getattr(da, "chunks", None)
    if chunks is None:
        return
    _, dim = grid.axes[ax_name]._get_position_name(da)
    if len(chunks[da.dims.index(dim)]) <= 1:
        return
    positions = {p for ps in list(sig.in_ax_positions) + list(sig.out_ax_positions) for p in ps}
    if positions & {"inner", "outer"}:
        raise NotImplementedError(
            "Cannot chunk along a core dimension for a grid ufunc which has a signature which "
            "includes one of the axis positions ['inner', 'outer']."
            "Consider rechunking to a single chunk along this dimension if possible."
        )


class Grid:
    """Multiple :class:`Axis` objects describing the staggered topology of a dataset."""

    def __init__(self, ds, coords: Optional[Mapping[str, Mapping[str, str]]] = None, fill_value=None,
                 default_shifts=None, padding=None, face_connections=None, metrics=None,
                 autoparse_metadata: bool = True, fuse: bool = False, **kwargs):
        if "boundary" in kwargs:
            raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
        given = ds
        if is_xarray(ds):
            ds = from_xarray(ds)
        if not isinstance(ds, Dataset):
            raise TypeError(f"ds argument to `xgcm.Grid` must be of type xarray.Dataset, but is of type {type(ds)}")
        # `_ds` is the dataset AS GIVEN (an xarray.Dataset stays one: code written against the reference reads
        # `grid._ds.<coord>`, its tests 69 times); the operators work on `_own_ds`, the library's view of it
        self._ds = given
        self._own_ds = ds
        # `fuse=True` (or `with grid.fused():`): diff / interp / min / max / derivative return deferred results whose
        # chains -- `(grid.diff(v, "X") - grid.diff(u, "Y")) / area` -- run as ONE fused kernel when the value is used
        # (xgcm_amd.lazy; the reference's own TODO, xgcm/grid.py:797-799).  Off by default: results are computed at the call.
        self._fuse = bool(fuse)
        self._fuse_local = threading.local()
        if autoparse_metadata:
            # COMODO attributes / SGRID topology of the dataset supply what the caller left out; what BOTH supply is a
            # conflict, never a silent choice (xgcm/grid.py:151-195 -- `coords` is always among the parsed kwargs, so
            # explicit `coords` need `autoparse_metadata=False` exactly as there)
            from .metadata import parse_metadata

            _, parsed = parse_metadata(ds)
            given = {"coords": coords, "fill_value": fill_value, "default_shifts": default_shifts, "padding": padding,
                     "face_connections": face_connections, "metrics": metrics}
            duplicates = [k for k in given if k in parsed and given[k] is not None]
            if duplicates:
                raise ValueError(
                    f"Autoparsed Grid kwargs: '{', '.join(duplicates)}' conflict with "
                    f"user-supplied kwargs. Run with 'autoparse_metadata=False', or "
                    f"autoparse and amend kwargs before calling Grid constructer."
                )
            coords = parsed.get("coords", coords)
        if "periodic" in kwargs:
            raise ValueError(
                "The `periodic` argument has been removed. Use "
                "`padding='periodic'` (per axis if needed, e.g. "
                "`padding={'X': 'periodic', 'Y': 'fill'}`) instead. "
                "Previously `periodic=False` corresponded to `padding='fill'`."
            )
        if kwargs:
            raise TypeError(
                f"Grid.__init__() got unexpected keyword argument(s): {', '.join(repr(k) for k in kwargs)}"
            )
        if fill_value:
            warnings.warn(
                "The default fill_value will be changed to nan (from 0.0 previously) "
                "in future versions. Provide `fill_value=0.0` to preserve previous behavior.",
                category=DeprecationWarning,
            )
        if coords is None:  # (a dataset whose metadata names no axis parses to {}: an empty Grid, as in the reference)
            raise ValueError(
                "Could not determine Axis names - please provide them in the coords kwarg "
                "or provide a dataset from which they can be parsed"
            )
        names = list(coords.keys())
        per_axis_padding = self._map_kwargs_over_axes(padding, axes=names)
        per_axis_shifts = self._map_kwargs_over_axes(default_shifts, axes=names)
        per_axis_fill = self._map_kwargs_over_axes(fill_value, axes=names)
        # only an axis the user marked periodic can be the seam of a north fold (grid.py:241-249)
        self._explicitly_periodic_axes = {ax for ax, p in per_axis_padding.items() if p == "periodic"}
        if face_connections:
            self._facedim = list(face_connections.keys())[0]
            self._face_connections = face_connections
        else:
            self._facedim = None
            self._face_connections = None
        self._connected_axes: set = set()
        self.axes: "OrderedDict[str, Axis]" = OrderedDict()
        for ax in names:
            self.axes[ax] = Axis(
                ds,
                ax,
                coords=coords[ax],
                default_shifts=per_axis_shifts.get(ax, None),
                padding=per_axis_padding.get(ax, None),
                fill_value=per_axis_fill.get(ax, None),
            )

        if face_connections is not None:
            self._assign_face_connections(face_connections)
        self._validate_folds()

        self._own_metrics: Dict[frozenset, List[DataArray]] = {}
        self._device_cache: Dict[int, Any] = {}
        self._metric_ids: set = set()
        if metrics is not None:
            for key, value in metrics.items():
                self.set_metrics(key, value)

    # ---- complex topologies (reference grid.py:334-470) --------------------------------------
    def _assign_face_connections(self, fc) -> None:
        """Check that every face link is mirrored by its neighbour and hand each axis its links."""
        if len(fc) > 1:
            raise ValueError("Only one face dimension is supported for now. Instead found %r" % repr(fc.keys()))
        facedim = list(fc.keys())[0]
        if facedim not in self._own_ds.dims:
            raise ValueError(
                f"Face dimension {facedim} does not exist in the dataset. Found {list(self._own_ds.dims)} instead"
            )
        face_links = fc[facedim]
        if facedim in self._own_ds.coords:
            valid_faces = set(np.asarray(self._own_ds[facedim].values).tolist())
        else:
            valid_faces = set(range(self._own_ds.dims[facedim]))
        per_axis: Dict[str, Dict[Any, Tuple]] = {}
        for fidx, axis_links_of_face in face_links.items():
            for axis, (link_left, link_right) in axis_links_of_face.items():
                checked = []
                # a link seen from its left end must come back from the neighbour's right end,
                # unless the connection is reversed
                for link, position in ((link_left, 1), (link_right, 0)):
                    if link is None:
                        checked.append(None)
                        continue
                    idx, ax, rev = link
                    back_position = int(not position) if rev else position
                    try:
                        neighbor_link = face_links[idx][ax][back_position]
                    except (KeyError, IndexError):
                        raise KeyError(
                            "Couldn't find a face link for face %r"
                            "in axis %r at position %r" % (idx, ax, back_position)
                        )
                    idx_n, ax_n, rev_n = neighbor_link
                    for name in (ax, ax_n):
                        if name not in self.axes:
                            raise KeyError("axis %r is not a valid axis" % name)
                    for face in (idx, idx_n):
                        if face not in valid_faces:
                            raise IndexError("%r is not a valid index for facedimension %r" % (face, facedim))
                    if (idx_n != fidx) or (ax_n != axis) or (rev_n != rev):
                        raise ValueError(
                            "Face link mismatch: neighbor doesn't"
                            " correctly link back to this face. "
                            "face: %r, axis: %r, position: %r, "
                            "rev: %r, link: %r, neighbor_link: %r"
                            % (fidx, axis, position, rev, link, neighbor_link)
                        )
                    checked.append((idx, self.axes[ax], rev))
                per_axis.setdefault(axis, {})[fidx] = tuple(checked)
        for axis, links in per_axis.items():
            self.axes[axis]._facedim = facedim
            self.axes[axis]._face_connections = links
        # axes that any link touches, as the linked edge or as the neighbour's axis
        self._connected_axes = set(per_axis)
        for links in face_links.values():
            for pair in links.values():
                self._connected_axes.update(link[1] for link in pair if link is not None)

    def _links_swap_axes(self, ax_name: str, first_face: bool = False):
        """does any face link on `ax_name` lead to ANOTHER axis of the neighbour (or any link of another axis lead here)?
        `first_face`: the lowest face number that has such a link (None if none) instead of a bool"""
        if self._face_connections is None:
            return None if first_face else False
        hits = []
        for faces in self._face_connections.values():
            for face, links in faces.items():
                for axis, pair in links.items():
                    for link in pair:
                        if link is not None and (axis == ax_name or link[1] == ax_name) and link[1] != axis:
                            hits.append(face)
        if first_face:
            return min(hits) if hits else None
        return bool(hits)

    def _validate_folds(self) -> None:
        """Resolve north-fold paddings: the seam is the one explicitly periodic other axis."""
        self._folds = {}
        for axname, axis in self.axes.items():
            spec = axis._padding
            if not FoldSpec.looks_like(spec):
                continue
            candidates = [o for o in self.axes if o != axname and o in self._explicitly_periodic_axes]
            if not candidates:
                raise ValueError(
                    f"A fold padding on axis {axname!r} requires an explicitly "
                    "periodic seam axis (the zonal wrap), but no other axis was "
                    "explicitly marked periodic. Set e.g. "
                    "padding={'X': 'periodic', '" + str(axname) + "': {'fold': ...}}."
                )
            if len(candidates) > 1:
                raise ValueError(
                    f"A fold padding on axis {axname!r} is ambiguous: more than one "
                    f"explicitly periodic axis could be the seam ({candidates}). "
                    "Multiple candidate seam axes are not supported."
                )
            self._folds[axname] = {"seam_axis": candidates[0], "pivot": spec["fold"], "south": spec["south"]}
        if self._folds and self._face_connections is not None:
            raise NotImplementedError(
                "Combining a north-fold boundary with face_connections is not "
                f"supported (fold axes: {sorted(self._folds)}). Use one or the "
                "other."
            )
        if self._folds:
            warnings.warn(
                "The north-fold (tripolar) boundary condition is experimental. "
                "Its API and numerical behavior may change in future releases, "
                "and it has not yet been validated across the full range of grid "
                "configurations. Please review results carefully and report any "
                "issues at https://github.com/xgcm/xgcm/issues.",
                category=UserWarning,
            )

    # ---- kwarg plumbing (reference grid.py:291-332) -----------------------------------------
    def _map_kwargs_over_axes(self, kwargs, axes: Optional[Iterable[str]] = None) -> Dict[str, Any]:
        if isinstance(kwargs, dict):
            return kwargs
        return {ax: kwargs for ax in (self.axes if axes is None else axes)}

    def _complete_user_kwargs_using_axis_defaults(self, user_kwargs, property: str) -> Dict[str, Any]:
        defaults = {ax: getattr(self.axes[ax], property) for ax in self.axes}
        if user_kwargs is None:
            return defaults
        return {**defaults, **self._map_kwargs_over_axes(user_kwargs)}

    def __repr__(self) -> str:
        lines = ["<xgcm.Grid>"]
        for name, axis in self.axes.items():
            lines.append("%s Axis (%s, padding=%r):" % (name, "periodic" if axis.periodic else "not periodic", axis.padding))
            lines += axis._coord_desc()
        return "\n".join(lines)

    # ---- deferred results (xgcm_amd.lazy) -----------------------------------------------------
    @contextlib.contextmanager
    def fused(self):
        """Inside the block the built-in 1-D operators return deferred results (`xgcm_amd.lazy.LazyArray`): chains of
        them and `+ - * /` are matched against the fused kernels when a value is first used -- bit-identical to the
        eager chain, one launch instead of three or four.  Per thread; nests."""
        depth = getattr(self._fuse_local, "depth", 0)
        self._fuse_local.depth = depth + 1
        _lazy.enter()
        try:
            yield self
        finally:
            _lazy.leave()
            self._fuse_local.depth = depth

    @property
    def _fusing(self) -> bool:
        return self._fuse or getattr(self._fuse_local, "depth", 0) > 0

    # ---- residency helpers ------------------------------------------------------------------
    def _resident(self, da: DataArray, like) -> DataArray:
        """Metric arrays of grid._own_ds are uploaded once and kept in HBM while HBM data is processed."""
        if _is_tensor(da.data) or not _is_tensor(like):
            return da
        key = id(da.data)
        if key not in self._metric_ids:  # temporaries (e.g. dx*dy products) are uploaded, not cached
            return da._replace(data=_dev.asdevice(da.data))
        hit = self._device_cache.get(key)
        if hit is None or hit[0] is not da.data:
            hit = (da.data, _dev.asdevice(da.data))
            self._device_cache[key] = hit
        return da._replace(data=hit[1])

    @staticmethod
    def _wrap_in(obj):
        """xarray objects -> xgcm_amd labelled objects; returns (obj, was_xarray)."""
        if isinstance(obj, dict):
            conv = {k: Grid._wrap_in(v) for k, v in obj.items()}
            return {k: v[0] for k, v in conv.items()}, any(v[1] for v in conv.values())
        if is_xarray(obj):
            return from_xarray(obj), True
        return obj, False

    # ---- metrics (reference grid.py:472-657) ------------------------------------------------
    @property
    def _metrics(self):
        """{frozenset(axes): [metric, ...]} in the container type of the dataset the grid was given (the reference keeps
        `ds[name].reset_coords(drop=True)` there and its tests read it); the operators use `_own_metrics`"""
        if isinstance(self._ds, Dataset):
            return self._own_metrics
        return {k: [to_xarray(m) for m in ms] for k, ms in self._own_metrics.items()}

    def set_metrics(self, key, value, overwrite: bool = False) -> None:
        metric_axes = frozenset(_maybe_promote_str_to_list(key))
        missing = [ma for ma in metric_axes if ma not in self.axes]
        if missing:
            raise KeyError(f"Metric axes {missing!r} not compatible with grid axes {tuple(self.axes)!r}")
        varnames = _maybe_promote_str_to_list(value)
        for v in varnames:
            if v not in self._own_ds:  # (membership only: a converted xarray.Dataset loads a variable when it is indexed)
                raise KeyError(f"Metric variable {v} not found in dataset.")
        if metric_axes in self._own_metrics:
            # NB the reference only considers the LAST name of `value` here (grid.py:488-512)
            new = self._own_ds[varnames[-1]].reset_coords(drop=True)
            replaced = False
            for i, old in enumerate(self._own_metrics[metric_axes]):
                if set(new.dims) == set(old.dims):
                    if not overwrite:
                        raise ValueError(
                            f"Metric variable {old.name} with dimensions {old.dims} already assigned in metrics."
                            f" Overwrite {old.name} with {varnames[-1]} by setting overwrite=True."
                        )
                    self._own_metrics[metric_axes][i] = new
                    replaced = True
            if not replaced:
                self._own_metrics[metric_axes].append(new)
        else:
            self._own_metrics[metric_axes] = [self._own_ds[v].reset_coords(drop=True) for v in varnames]
        self._metric_ids = {id(m.data) for ms in self._own_metrics.values() for m in ms}

    def _get_dims_from_axis(self, da, axis) -> List[str]:
        da = _maybe_unpack_vector_component(da)
        dims = []
        for ax in _maybe_promote_str_to_list(axis):
            if ax not in self.axes:
                raise KeyError(f"Did not find axis {ax} from data array {da.name}")
            found = [d for d in self.axes[ax].coords.values() if d in da.dims]
            if len(found) != 1:
                raise ValueError(
                    f"Did not find single matching dimension {da.dims} from {da.name} corresponding to axis {ax}, got {found}."
                )
            dims.append(found[0])
        return dims

    def get_metric(self, array, axes, _layout=None, _factors=False):
        """Metric that broadcasts against `array` for the given axes (conditions 1-4 of the reference).

        `_layout` (internal, the operators pass the dims of the array they weight): a metric that has to be formed
        as a PRODUCT of registered metrics (a volume = area(Y,X) * thickness(Z)) is laid out with its dims in that
        order.  xarray's order -- first factor's dims, then the new ones: (Y, X, Z) -- made every use a transposed
        5 GB copy: `integrate(T, [X, Y, Z])` took 16 ms.  Same factors in the same order, same values."""
        if is_xarray(array):  # xarray in -> xarray out, like every other method (the reference hands back `grid._ds`'s own)
            return to_xarray(self.get_metric(from_xarray(array), axes))

        def product(parts):
            if _factors:  # internal: the factors themselves (separable weights are applied stage by stage)
                return tuple(parts)
            if _layout is None:
                return functools.reduce(operator.mul, parts[1:], parts[0])
            acc = parts[0]
            for p in parts[1:]:
                acc = acc._binary(p, "mul", dims_order=tuple(_layout))
            return acc

        array_dims = set(array.dims)
        self._get_dims_from_axis(array, frozenset(axes))
        registered = set(tuple(k) for k in self._own_metrics.keys())
        wanted = set(itertools.permutations(tuple(axes)))
        exact = registered.intersection(wanted)
        found = None
        if exact:
            candidates = self._own_metrics[frozenset(*exact)]
            for mv in candidates:  # (1) registered under these axes at this position
                if set(mv.dims).issubset(array_dims):
                    found = mv
                    break
            if found is None:  # (2) registered under these axes elsewhere: interpolate it
                mv = candidates[-1]
                warnings.warn(
                    f"Metric at {array.dims} being interpolated from metrics at dimensions {mv.dims}. Boundary value set to 'extend'."
                )
                found = self.interp_like(mv, array, "extend", None)
        else:
            fallback = None
            fallback_locked = False
            for combo in iterate_axis_combinations(axes):
                try:
                    pools = [self._own_metrics[ac] for ac in combo]
                except KeyError:
                    continue
                for pick in itertools.product(*pools):
                    if set(d for mv in pick for d in mv.dims).issubset(array_dims):
                        found = product(pick)  # (3) product of sub-axis metrics
                        break
                    if not fallback_locked:
                        fallback = pick
                if found is not None:
                    break
                fallback_locked = True
            if found is None and fallback is not None:  # (4) interpolate the fallback combination
                warnings.warn(
                    f"Metric at {array.dims} being interpolated from metrics at dimensions {[pc.dims for pc in fallback]}. Boundary value set to 'extend'."
                )
                parts = [self.interp_like(pc, array, "extend", None) for pc in fallback]
                found = product(parts)
        if found is None:
            raise KeyError(f"Unable to find any combinations of metrics for array dims {array_dims!r} and axes {axes!r}")
        if _factors and not isinstance(found, tuple):
            found = (found,)
        return found

    def interp_like(self, array, like, padding=None, fill_value=None, **kwargs):
        """Interpolate `array` onto the positions of `like` wherever they differ (grid.py:659-716)."""
        if "boundary" in kwargs:
            raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
        shifted = []
        for name, axis in self.axes.items():
            try:
                here, _ = axis._get_position_name(array)
                there, _ = axis._get_position_name(like)
            except KeyError:
                continue
            if here != there:
                shifted.append(name)
        # like the reference (grid.py:710-715) the target position is NOT passed on: each axis moves by
        # its default shift, which is `like`'s position on the usual two-position (center + one edge) axes
        out = self._1d_grid_ufunc_dispatch("interp", array, shifted, fill_value=fill_value, padding=padding)
        if is_xarray(like) and isinstance(out, DataArray):  # (a registered metric of ours interpolated like an xarray field)
            out = to_xarray(out)
        return out

    # ---- the hot path: 1-D operators ---------------------------------------------------------
    def _create_1d_grid_ufunc_signatures(self, da, axis, to) -> List[_GridUFuncSignature]:
        sigs = []
        for ax_name in axis:
            ax = self.axes[ax_name]
            from_pos, _ = ax._get_position_name(da)
            to_pos = to[ax_name]
            if to_pos is None:
                to_pos = ax._default_shifts[from_pos]
            sigs.append(_GridUFuncSignature.from_string(f"({ax_name}:{from_pos})->({ax_name}:{to_pos})"))
        return sigs

    def _1d_grid_ufunc_dispatch(self, funcname, data, axis, to=None, metric_weighted=None, other_component=None,
                                _divide_by=None, **kwargs):
        """Apply the matching built-in 1-D grid ufunc along each axis in turn (grid.py:728-836).

        `_divide_by` (internal, used by `derivative`): axes tuple whose metric at the OUTPUT position
        divides the result of the (single) axis inside the same kernel launch."""
        if "keep_coords" in kwargs:
            raise ValueError(
                "The 'keep_coords' argument has been removed. Coordinates compatible with the output are now always preserved."
            )
        data, was_xr = self._wrap_in(data)
        if isinstance(axis, str):
            axis = [axis]
        data = _check_data_input(data, self)
        first = _maybe_unpack_vector_component(data)
        chunked = getattr(first, "chunks", None) is not None
        if chunked and (isinstance(data, dict) or other_component is not None or any(gridops.complex_topology(self, a) for a in axis)):
            raise NotImplementedError(CHUNKED_INPUT_MESSAGE)  # (halos from other faces / partner components: whole arrays)
        to = self._map_kwargs_over_axes(to)
        if isinstance(metric_weighted, str):
            metric_weighted = (metric_weighted,)
        metric_weighted = self._map_kwargs_over_axes(metric_weighted)
        signatures = self._create_1d_grid_ufunc_signatures(first, axis=axis, to=to)

        array = first
        vector_key = next(iter(data)) if isinstance(data, dict) else None
        steps = list(zip(signatures, axis))
        i = 0
        while i < len(steps):
            sig, ax_name = steps[i]
            if (i + 1 < len(steps) and vector_key is None and other_component is None and _divide_by is None
                    and not (self._fusing and isinstance(array, _lazy.LazyArray))):
                w_a = metric_weighted.get(ax_name) if isinstance(metric_weighted, dict) else None
                w_b = metric_weighted.get(steps[i + 1][1]) if isinstance(metric_weighted, dict) else None
                fused = None
                if not w_a and not w_b:
                    fused = self._two_axes_in_one_pass(funcname, array, steps[i], steps[i + 1], kwargs)
                elif w_a and w_b and tuple(w_a) == tuple(w_b):  # metric_weighted=("X", "Y") on both axes: three metric planes
                    fused = self._two_axes_in_one_pass(funcname, array, steps[i], steps[i + 1], kwargs, weighted=tuple(w_a))
                if fused is not None:
                    array = fused
                    i += 2
                    continue
            i += 1
            ufunc, remaining = _select_grid_ufunc(funcname, sig, module=gridops, **kwargs)
            if chunked and funcname != "cumsum":
                _refuse_chunked_core_dim_with_changing_length(self, first, ax_name, sig)
            weighted = metric_weighted.get(ax_name) if isinstance(metric_weighted, dict) else None
            out_dims = _shifted_dims(self, array, ax_name, sig.out_ax_positions[0][0], sig.in_ax_positions[0][0])
            m_in = m_out = None
            post_divide = late_error = None
            if weighted:
                m_in = self._resident(self.get_metric(array, weighted, _layout=array.dims), _lazy.like(array))
                if any(d not in array.dims for d in m_in.dims):
                    # the metric found (interpolated by default shifts, so not always onto the array's points) carries a
                    # dim the array lacks: the reference's `array * metric` broadcasts by name into an outer product
                    # BEFORE the operator (xgcm/grid.py:804-808) -- the explicit product does the same
                    array = _lazy.plain(array) * m_in
                    m_in = None
                    out_dims = _shifted_dims(self, array, ax_name, sig.out_ax_positions[0][0], sig.in_ax_positions[0][0])
                try:
                    m_out = self._resident(self.get_metric(_DimsOnly(out_dims, array.name), weighted, _layout=out_dims), _lazy.like(array))
                except (KeyError, ValueError) as exc:
                    late_error = exc  # the reference looks this metric up AFTER the operator ran (:829-831): its errors first
                if m_out is not None and any(d not in out_dims for d in m_out.dims):
                    post_divide, m_out = m_out, None  # ... and `array / metric` AFTER it, likewise by name
            if _divide_by is not None and late_error is None:
                try:
                    dx = self._resident(self.get_metric(_DimsOnly(out_dims, array.name), _divide_by, _layout=out_dims), _lazy.like(array))
                except (KeyError, ValueError) as exc:
                    late_error, dx = exc, None  # `grid.diff(...)` first, then the metric (xgcm/grid.py:1576-1578)
                if dx is None:
                    pass
                elif any(d not in out_dims for d in dx.dims):
                    # the metric found for the result does not live on the result's points (drC on Zp1 interpolated to Z
                    # for a difference that went to Zl ...): the reference's `diff / dx` then BROADCASTS by name into an
                    # outer product (xgcm/grid.py:1576-1578) -- the explicit quotient does the same
                    post_divide = dx
                elif m_out is None:
                    m_out = dx
                else:
                    post_divide = dx  # two successive divisions cannot be merged bit-exactly
            if sum(d in array.dims for d in self.axes[ax_name].coords.values()) > 1 and any(
                    any(w) for w in (getattr(ufunc, "padding_width", None) or {}).values()):
                # a metric product (this step's or an earlier one's) gave the array a second dim of this axis: the
                # reference's pad looks the axis' dim up again and refuses (xgcm/padding.py:595); an operator that pads
                # nothing goes on with the dim its signature names
                self.axes[ax_name]._get_position_name(array)
            arg = {vector_key: array} if vector_key is not None else array
            if isinstance(ufunc, gridops.HipGridUFunc):
                if m_in is not None and gridops.complex_topology(self, ax_name) and (vector_key is not None or other_component is not None):
                    # halos come from other faces / the folded row and must be halos of the PRODUCT (the reference
                    # multiplies first, grid.py:804-808).  A scalar field takes them as the product of two halo slabs and
                    # stays in ONE pass (gridops._fused); a vector component's halos mix components: explicit product
                    # pass, then the halo-fused kernel with the output metric
                    array = array * m_in
                    arg = {vector_key: array} if vector_key is not None else array
                    m_in = None
                if self._fusing and post_divide is None and late_error is None:
                    deferred = _lazy.defer_stencil(self, funcname, ufunc, sig, arg, ax_name, other_component, m_in, m_out,
                                                   remaining, out_dims)
                    if deferred is not None:
                        array = deferred
                        continue
                array = ufunc(self, _lazy.plain(arg), axis=[(ax_name,)], other_component=_lazy.plain(other_component),
                              metric_in=m_in, metric_out=m_out, **remaining)
            else:  # a plain GridUFunc registered in gridops (none of the built-ins): unfused sequence
                if m_in is not None:
                    array = array * m_in
                    arg = {vector_key: array} if vector_key is not None else array
                array = ufunc(self, arg, axis=[(ax_name,)], other_component=other_component, **remaining)
                if m_out is not None:
                    array = array / m_out
            if late_error is not None:
                raise late_error
            if (m_in is not None or m_out is not None) and array.name is not None:
                array = array._replace(name=_name_after(array.name, m_in, m_out))
            if post_divide is not None:
                array = array / post_divide
        if was_xr and isinstance(array, _lazy.LazyArray) and array.is_deferred:
            # deferred: converting now would evaluate it.  The result stays a LazyArray -- it combines with xarray objects
            # through `+ - * /` like the arrays it came from -- and `.compute()` / `.to_xarray()` hand over the
            # xarray.DataArray, as `.compute()` of a dask-backed result does there
            array._xr = True
            return array
        return to_xarray(array) if was_xr else array

    def _two_axes_in_one_pass(self, funcname, array, step_a, step_b, kwargs, weighted=None):
        """`op` along two axes that are the LAST TWO dims of `array` in one kernel launch
        (xg_stencil2d_f64): same bits as the two sequential passes of the reference loop
        (xgcm/grid.py:800-832, whose TODO at :798-800 asks for exactly this), half the traffic.
        Returns None when the pair does not qualify; the caller then runs the axes one by one."""
        if funcname not in ("diff", "interp", "min", "max") or set(kwargs) - {"padding", "fill_value"}:
            return None
        if array.ndim < 2 or getattr(array, "chunks", None) is not None:
            return None
        if gridops.is_integer_data(array.data):
            return None  # integers: one axis at a time on int64 lanes (interp leaves the integer domain between the axes)
        if _dt.np_dtype(array.data) == np.float16:
            return None  # float16 is a storage type: numpy rounds to it BETWEEN the axes -- one axis at a time (found by the GPU sweep)
        (sig_a, ax_a), (sig_b, ax_b) = step_a, step_b
        if ax_a == ax_b or gridops.complex_topology(self, ax_a) or gridops.complex_topology(self, ax_b):
            return None  # halos from other faces / the folded row: one axis at a time (xg_stencil1d_halo)
        try:
            ufa, _ = _select_grid_ufunc(funcname, sig_a, module=gridops)
            ufb, _ = _select_grid_ufunc(funcname, sig_b, module=gridops)
        except NotImplementedError:
            return None  # (one axis at a time, so that the error -- if any -- is the one of the FIRST bad axis, as there)
        if not (isinstance(ufa, gridops.HipGridUFunc) and isinstance(ufb, gridops.HipGridUFunc)):
            return None
        try:
            dim_a = self.axes[ax_a].coords[ufa.from_pos]
            dim_b = self.axes[ax_b].coords[ufb.from_pos]
            out_a = self.axes[ax_a].coords[ufa.to_pos]
            out_b = self.axes[ax_b].coords[ufb.to_pos]
        except KeyError:
            return None
        if {dim_a, dim_b} != set(array.dims[-2:]):
            return None
        pad_a = tuple(next(iter(ufa.padding_width.values())))
        pad_b = tuple(next(iter(ufb.padding_width.values())))
        bc = self._complete_user_kwargs_using_axis_defaults(kwargs.get("padding"), "padding")
        fv = self._complete_user_kwargs_using_axis_defaults(kwargs.get("fill_value"), "fill_value")
        if any(not isinstance(bc[a], str) for a in (ax_a, ax_b)):
            return None  # missing boundary condition / fold spec: the sequential path raises the right error
        x_is_a = array.dims[-1] == dim_a
        padx, pady = (pad_a, pad_b) if x_is_a else (pad_b, pad_a)
        ax_x, ax_y = (ax_a, ax_b) if x_is_a else (ax_b, ax_a)
        if not _dev.stencil2d_supported(array.data, padx, pady):
            return None
        host = not _is_tensor(array.data)
        planes = None
        named = array.name
        if weighted is not None:
            # the metric at the input positions, between the two axes and at the output positions (xgcm/grid.py:804-828
            # multiplies before and divides after EACH axis); each must be a plain (Y, X) plane over the last two dims
            dims_mid = tuple(out_a if d == dim_a else d for d in array.dims)
            dims_out = tuple(out_b if d == dim_b else d for d in dims_mid)
            planes = []
            for dims in (array.dims, dims_mid, dims_out):
                try:
                    m = self.get_metric(_DimsOnly(dims, array.name), weighted, _layout=dims)
                except (KeyError, ValueError):
                    return None
                if tuple(m.dims) != tuple(dims[-2:]) or tuple(m.shape) != tuple(array.shape[-2:]):
                    return None  # broadcast / extra dims: the per-axis kernels take any metric pattern
                if _dt.np_dtype(m.data) != _dt.np_dtype(array.data):
                    return None  # mixed precision (a float32 field, float64 metrics): numpy's steps, one axis at a time
                planes.append(self._resident(m, array.data).data)
                named = _name_after(named, m)
        out = _dev.stencil2d(funcname, array.data, 0 if x_is_a else 1, padx, bc[ax_x], float(fv[ax_x] or 0.0), pady,
                             bc[ax_y], float(fv[ax_y] or 0.0), **({"metrics": planes} if planes is not None else {}))
        rename = {dim_a: out_a, dim_b: out_b}
        res = DataArray(_dev.tohost(out) if host else out, tuple(rename.get(d, d) for d in array.dims), name=named)
        return _reattach_coords([res], self, None, {out_a, out_b}, [array])[0]

    def interp(self, da, axis, **kwargs):
        """Interpolate neighboring points to the intermediate grid point along this axis."""
        return self._1d_grid_ufunc_dispatch("interp", da, axis, **kwargs)

    def diff(self, da, axis, **kwargs):
        """Difference neighboring points to the intermediate grid point."""
        return self._1d_grid_ufunc_dispatch("diff", da, axis, **kwargs)

    def min(self, da, axis, **kwargs):
        """Minimum of neighboring points on the intermediate grid point."""
        return self._1d_grid_ufunc_dispatch("min", da, axis, **kwargs)

    def max(self, da, axis, **kwargs):
        """Maximum of neighboring points on the intermediate grid point."""
        return self._1d_grid_ufunc_dispatch("max", da, axis, **kwargs)

    def _apply_vector_function(self, function, vector, **kwargs):
        """Each component of a C-grid vector moved to the cell centre with the other one as its
        `other_component` (reference grid.py:1420-1471; deprecated there, kept for its tests)."""
        if not (isinstance(vector, dict) and len(vector) == 2):
            raise ValueError(
                "Input is expected to be a dictionary with two key/value pairs which map grid axis to the vector component parallel to that axis"
            )
        warnings.warn(
            "`interp_2d_vector` and `diff_2d_vector` will be removed from future releases."
            "The same functionality will be accessible under the `xgcm.Grid.diff` and `xgcm.Grid.interp` methods, please see those docstrings for details.",
            category=DeprecationWarning,
        )
        to = kwargs.get("to", "center")
        if to != "center":
            raise NotImplementedError("Only vector interpolation to cell center is implemented, but got to=%r" % to)
        for axis_name, component in vector.items():
            position, _ = self.axes[axis_name]._get_position_name(self._wrap_in(component)[0])
            if position == "center":
                raise NotImplementedError(
                    "Only vector interpolation to cell "
                    "center is implemented, but vector "
                    "%s component is defined at center "
                    "(dims: %r)" % (axis_name, tuple(component.dims))
                )
        x_name, y_name = list(vector)
        x_component = function({x_name: vector[x_name]}, x_name, other_component={y_name: vector[y_name]}, **kwargs)
        y_component = function({y_name: vector[y_name]}, y_name, other_component={x_name: vector[x_name]}, **kwargs)
        return {x_name: x_component, y_name: y_component}

    def diff_2d_vector(self, vector, **kwargs):
        """Difference a 2D vector to the intermediate grid point (complex topologies)."""
        return self._apply_vector_function(self.diff, vector, **kwargs)

    def interp_2d_vector(self, vector, **kwargs):
        """Interpolate a 2D vector to the intermediate grid point (complex topologies)."""
        return self._apply_vector_function(self.interp, vector, **kwargs)

    def derivative(self, da, axis, **kwargs):
        """Centered-difference derivative: `diff(da, axis) / get_metric(diff, (axis,))` (grid.py:1576-1578)."""
        return self._1d_grid_ufunc_dispatch("diff", da, axis, _divide_by=(axis,), **kwargs)

    def cumsum(self, da, axis, to=None, padding=None, fill_value=None, metric_weighted=None, reverse=False, **kwargs):
        """Cumulative sum along `axis`, shifted to the neighbouring position (grid.py:1183-1418)."""
        if "boundary" in kwargs:
            raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
        if "keep_coords" in kwargs:
            raise ValueError(
                "The 'keep_coords' argument has been removed. Coordinates compatible with the output are now always preserved."
            )
        pre_weight = kwargs.pop("_pre_weight", None)  # cumint: `da * metric` folded into the first scan's load
        if kwargs:
            raise TypeError(f"cumsum() got unexpected keyword argument(s): {list(kwargs)}")
        da, was_xr = self._wrap_in(da)
        if isinstance(axis, str):
            axis = [axis]
        to = self._map_kwargs_over_axes(to)
        if isinstance(reverse, dict):
            extra = [a for a in reverse if a not in axis]
            if extra:
                raise ValueError(
                    f"`reverse` was given for axes {extra} which are not being "
                    f"cumulatively summed (axis={axis}). Only pass `reverse` for "
                    f"the axes in `axis`."
                )
        reverse = self._map_kwargs_over_axes(reverse)
        if isinstance(metric_weighted, str):
            metric_weighted = (metric_weighted,)
        metric_weighted = self._map_kwargs_over_axes(metric_weighted)
        all_padding = self._complete_user_kwargs_using_axis_defaults(padding, "padding")
        all_fill = self._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")

        data = da
        for ax in [self.axes[name] for name in axis]:
            pos, dim = ax._get_position_name(da)
            rev = bool(reverse.get(ax.name, False))
            weighted = metric_weighted.get(ax.name) if isinstance(metric_weighted, dict) else None
            # (the steps below come in the reference's order, xgcm/grid.py:1303-1413, so that a call with several things
            # wrong raises the error the reference raises: input metric, trim table, pad, target dim, output metric)
            m_in = m_out = None
            weight_names = []  # the DataArrays whose products / quotients decide the result's name (xarray's rule)
            generic_pad = gridops.complex_topology(self, ax.name)
            if pre_weight is not None:
                if weighted or generic_pad:  # two input factors / pad-after route: explicit product first
                    data = data * pre_weight
                else:
                    m_in = _aligned_view(pre_weight, data.dims)
                    weight_names.append(pre_weight)
                pre_weight = None
            two_axis_dims = False
            if weighted:
                w_in = self.get_metric(data, weighted, _layout=data.dims)
                if any(d not in data.dims for d in w_in.dims):
                    # a metric that does not live on the field's points (found by default-shift interpolation): the
                    # reference's `data * metric` is an outer product by name (:1311-1313); its pad then looks the axis'
                    # dim up again and refuses an array that now has two of them (xgcm/padding.py:595)
                    data = data * self._resident(w_in, data.data)
                    two_axis_dims = True
                else:
                    weight_names.append(w_in)
                    m_in = _aligned_view(self._resident(w_in, data.data), data.dims)
            ax_to = to.get(ax.name) if isinstance(to, dict) else None
            if ax_to is None:
                ax_to = ax._default_shifts[pos]
            trim_lo, trim_hi, pad_lo, pad_hi = _cumsum_trim_pad(pos, ax_to, rev, ax)
            bc = all_padding[ax.name]
            if pad_lo or pad_hi:
                if two_axis_dims or sum(d in data.dims for d in ax.coords.values()) > 1:  # (this step's product or an earlier one's)
                    ax._get_position_name(data)
                if bc is None and not generic_pad:
                    raise no_boundary_error(ax.name)
                if not generic_pad and self._face_connections is not None and self._facedim in data.dims:
                    # (an axis no link touches: the reference's pad still walks the faces, xgcm/padding.py:394-396)
                    for i in range(data.sizes[self._facedim]):
                        if i not in self._face_connections[self._facedim]:
                            raise KeyError(i)
                if generic_pad:
                    # the topology's own refusals (a face the connections leave out, an open edge without a boundary
                    # condition) come from the reference's pad, i.e. BEFORE the target dim is looked up: run the checks now
                    swapping = self._links_swap_axes(ax.name, first_face=True) if (trim_lo or trim_hi) else None
                    trimmed_faces_do_not_fit = ValueError(
                        # the reference pads the TRIMMED cumulative field through the topology (xgcm/grid.py:1385-1395): across
                        # a link that swaps axes the trimmed faces (n x (n - 1)) no longer fit each other and its concat fails
                        f"cumsum along {ax.name!r} from {pos!r} to {ax_to!r} trims the field along an axis whose face "
                        "connections swap axes: the trimmed faces are no longer square and cannot exchange halos")
                    try:
                        pad(data, self, {ax.name: (pad_lo, pad_hi)}, padding=padding, fill_value=fill_value, _dry=True)
                    except KeyError as missing_face:
                        # (it walks the faces by number: the concat of a lower face fails before a missing one is looked up)
                        if swapping is not None and missing_face.args and isinstance(missing_face.args[0], int) \
                                and swapping < missing_face.args[0]:
                            raise trimmed_faces_do_not_fit from None
                        raise
                    if swapping is not None:
                        raise trimmed_faces_do_not_fit
            fv = all_fill[ax.name]
            new_dim = ax.coords[ax_to]
            out_dims = tuple(new_dim if d == dim else d for d in data.dims)
            post_divide = None
            if weighted:
                w_out = self.get_metric(_DimsOnly(out_dims, data.name), weighted, _layout=out_dims)
                if any(d not in out_dims for d in w_out.dims):
                    post_divide = w_out  # `reattached / metric` (:1411-1413), by name as well
                else:
                    weight_names.append(w_out)
                    m_out = _aligned_view(self._resident(w_out, data.data), out_dims)
            num = data.get_axis_num(dim)
            host = not _is_tensor(data.data)
            # xarray's DataArray.cumsum skips NaN for floats (numpy.nancumsum); see DESIGN.md "unpinned"
            if generic_pad:
                # complex topology: the reference pads the TRIMMED cumulative field through the topology (grid.py:1389-1395),
                # i.e. its halo cells are the neighbouring faces' (the folded row's) cumulative edge values
                if pad_lo or pad_hi:
                    # one pass: the scan writes the padded layout (halo cells hold a placeholder), the halo cells are
                    # gathered from that very buffer -- a (pad_lo + pad_hi)-wide slab -- and put in place: no padded copy
                    src = data.data
                    if host and getattr(src, "nbytes", 0) >= getattr(_dev, "HOST_STREAM_MIN_BYTES", float("inf")):
                        src = _dev.asdevice(src)  # (the block-streamed host route hands back host blocks: keep this one in HBM)
                    out = _dev.cumsum1d(src, num, trim_lo, trim_hi, pad_lo, pad_hi, "fill", 0, rev, True, m_in, None)
                    buf = DataArray(out, data.dims, name=data.name)
                    halo = halo_cells(InteriorOf(buf, dim, pad_lo, pad_hi), self, ax.name, (pad_lo, pad_hi), padding=padding,
                                      fill_value=fill_value)
                    if tuple(halo.shape) != tuple(n if i != num else pad_lo + pad_hi for i, n in enumerate(out.shape)):
                        # links that swap axes need square faces, and trimming the cumulative field along one of them
                        # (center -> left ...) breaks that: the reference's pad of the trimmed field is ill-defined too
                        raise ValueError(
                            f"cumsum along {ax.name!r} from {pos!r} to {ax_to!r} trims the field along an axis whose face "
                            "connections swap axes: the trimmed faces are no longer square and cannot exchange halos")
                    out = _dev.put_halo(out, halo.data, num, pad_lo, pad_hi)
                else:
                    out = _dev.cumsum1d(data.data, num, trim_lo, trim_hi, 0, 0, None, 0.0, rev, True, m_in, None)
                res = DataArray(_dev.tohost(out) if host else out, out_dims, name=data.name)
                if weighted:
                    res = res / self._resident(self.get_metric(res, weighted, _layout=res.dims), res.data)

            else:
                # integer data: numpy.cumsum's int64 / uint64 accumulator and numpy.pad's cast of the fill value, both in
                # the device layer (xgcm_amd.dtypes)
                out = _dev.cumsum1d(data.data, num, trim_lo, trim_hi, pad_lo, pad_hi,
                                    bc if (pad_lo or pad_hi) else None, 0.0 if fv is None else fv, rev, True, m_in, m_out)
                res = DataArray(_dev.tohost(out) if host else out, out_dims, name=data.name)
            if (m_in is not None or m_out is not None) and res.name is not None:
                res = res._replace(name=_name_after(res.name, *weight_names))
            if (pad_lo or pad_hi) and self._facedim is not None and self._facedim in res.dims and res.dims[0] != self._facedim:
                # on a grid with face connections the reference pads EVERY axis through `_pad_face_connections`, which
                # rebuilds the array with `xr.concat(faces, dim=facedim)` -- the face dim comes out FIRST -- and `Grid.cumsum`
                # (unlike diff / interp) does not restore the input's order (xgcm/padding.py:540-572, xgcm/grid.py:1385-1395):
                # the same dims order here, as a view (no copy)
                res = res.transpose(self._facedim, *[d for d in res.dims if d != self._facedim])
            data = _reattach_coords([res], self, {ax.name: (pad_lo, pad_hi)}, {new_dim}, [data])[0]
            if post_divide is not None:
                data = data / self._resident(post_divide, data.data)
        return to_xarray(data) if was_xr else data

    def integrate(self, da, axis, **kwargs):
        """Finite-volume integral `(da * metric).sum(dims)` along one or more axes (grid.py:1580-1605)."""
        da, was_xr = self._wrap_in(da)
        skipna = kwargs.pop("skipna", None)
        keep_attrs = kwargs.pop("keep_attrs", False)
        if kwargs:
            raise TypeError(f"sum() got unexpected keyword argument(s): {list(kwargs)}")
        dims = self._get_dims_from_axis(da, axis)
        skip = True if skipna is None else bool(skipna)
        factors = self._weight_factors(da, axis, dims)
        if factors is None:  # the weight adds dims to `da`: the explicit product (broadcast result)
            weight = self._resident(self.get_metric(da, axis, _layout=da.dims), da.data)
            out = (da * weight).sum(dims, skipna=skip, keep_attrs=keep_attrs)
        else:
            # (`keep_attrs` keeps the attrs of what is summed -- the PRODUCT `da * metric`, which has none: xarray's binary
            # operators drop them; the reference's results carry no attrs either way)
            out = self._weighted_reduce(da, factors, dims, skip, False)
            # `da * weight` of the reference: nameless unless the weight carries the field's name (xarray's rule); a weight
            # that is itself a product of several metrics has no name
            weight_name = factors[0].name if len(factors) == 1 else None
            out = out._replace(name=da.name if weight_name == da.name else None)
        return to_xarray(out) if was_xr else out

    def _weight_factors(self, da, axis, dims):
        """The metric of `axis` for `da` as a list of FACTORS whose product it is (one factor unless the grid has
        to combine metrics, e.g. a volume = area(Y, X) * thickness(Z)); None when it has dims `da` lacks.
        Integrals over several axes apply each factor at the reduction that removes its first dim instead of
        materialising the product: `integrate(T, [X, Y, Z])` reads T once (8 B/cell), the product form moved a 3-D
        weight array as large as T as well.  The factor order of the reference is kept inside each stage; across
        stages the products associate differently -- multi-axis sums are tolerance results (1e-12) either way."""
        parts = [self._resident(f, da.data) for f in self.get_metric(da, axis, _factors=True)]
        if any(d not in da.dims for f in parts for d in f.dims):
            return None
        if any(not (set(f.dims) & set(dims)) for f in parts):  # a factor that none of the reductions removes
            parts = [functools.reduce(lambda a, b: a._binary(b, "mul", dims_order=tuple(da.dims)), parts[1:], parts[0])]
        return parts

    def _weighted_reduce(self, da, weight, dims, mode, keep_attrs=False):
        """sum over `dims` of da * weight, one `xg_reduce1d` launch per dim; `weight`: a DataArray or a list of factors,
        each riding in the launch that removes its first dim.  `mode`: skipna True / False; "valid" / "all" = the
        weights of the valid / of all cells of `da`; "mean_valid" / "mean_all" = the weighted mean (one dim: divided
        inside the kernel; several dims: numerator and denominator sums side by side, divided at the end)."""
        host = not _is_tensor(da.data)
        cur_dims = list(da.dims)
        data = da.data
        factors = list(weight) if isinstance(weight, (list, tuple)) else [weight]
        mean = mode in ("mean_valid", "mean_all")
        pair = mean and len(dims) > 1
        lead = []  # the leading pair dim once numerator / denominator travel together
        # a factor that has NONE of the summed dims (a metric registered under an axis it does not run along) still weights
        # every cell: it rides in the first launch, broadcast along the summed dim -- `(da * w).sum(d)`, not `w * da.sum(d)`
        loose = [f for f in factors if not any(d in f.dims for d in dims)]
        if factors and not loose and not any(dims[0] in f.dims for f in factors):
            # mixed precision with an unweighted first stage (a float32 field, float64 metrics that lack the first summed
            # dim): numpy has promoted the PRODUCT before it sums anything -- the field is widened first (x * 1.0 is exact)
            wide = np.result_type(_dt.np_dtype(data), *[_dt.np_dtype(f.data) for f in factors])
            if wide != _dt.np_dtype(data) and wide.kind == "f" and _dt.np_dtype(data).kind == "f":
                data = (da * wide.type(1.0)).data
        for i, d in enumerate(dims):
            now = [f for f in factors if d in f.dims] + (loose if i == 0 else [])
            factors = [f for f in factors if d not in f.dims and not any(f is x for x in loose)]
            w = None
            if now:
                wf = functools.reduce(lambda a, b: a._binary(b, "mul", dims_order=tuple(cur_dims)), now[1:], now[0])
                w = _aligned_view(wf, cur_dims)
                if lead:
                    w = w[None]
            num = len(lead) + cur_dims.index(d)
            if i == 0:
                step_mode = ({"mean_valid": "pair_valid", "mean_all": "pair_all"}[mode] if pair else mode)
            else:
                step_mode = mode if isinstance(mode, bool) else (mode == "mean_valid" if mean else False)
            data = _dev.reduce1d(data, num, w, step_mode)
            cur_dims.remove(d)
            if i == 0 and pair:
                lead = ["__pair__"]
        if pair:
            data = _dev.binary("div", data[0], data[1])
        coords = OrderedDict((k, c) for k, c in da.coords.items() if all(cd in cur_dims for cd in c.dims))
        return DataArray(_dev.tohost(data) if host else data, cur_dims, coords=coords, name=da.name,
                         attrs=da.attrs if keep_attrs else None)

    def cumint(self, da, axis, **kwargs):
        """Cumulative integral `cumsum(da * metric, axis)` (grid.py:1607-1660)."""
        da, was_xr = self._wrap_in(da)
        weight = self._resident(self.get_metric(da, axis, _layout=da.dims), da.data)
        if [d for d in weight.dims if d not in da.dims] or gridops.is_integer_data(da.data):
            res = self.cumsum(da * weight, axis, **kwargs)  # the product has more dims than `da` / integer data
        else:
            res = self.cumsum(da, axis, _pre_weight=weight, **kwargs)  # same products, formed inside the scan
        return to_xarray(res) if was_xr else res

    def average(self, da, axis, **kwargs):
        """Metric-weighted mean `sum(da*w) / sum(w where da valid)` (grid.py:1662-1685)."""
        da, was_xr = self._wrap_in(da)
        kwargs.pop("keep_attrs", None)
        skipna = kwargs.pop("skipna", None)
        if kwargs:
            raise TypeError(f"mean() got unexpected keyword argument(s): {list(kwargs)}")
        dims = self._get_dims_from_axis(da, axis)
        skip = True if skipna is None else bool(skipna)
        factors = self._weight_factors(da, axis, dims)
        if factors is None:  # weight adds dims: explicit broadcast products
            weight = self._resident(self.get_metric(da, axis, _layout=da.dims), da.data)
            num = (da * weight).sum(dims, skipna=skip)
            ones = _valid_mask(da) if skip else da._replace(data=_ones_like(da.data), coords=OrderedDict())
            den = (ones * weight).sum(dims, skipna=False)
            out = (num / den)._replace(name=da.name)  # (`da.weighted(w).mean(...)` keeps the array's name)
        else:
            # ONE pass over `da` (8 B/cell): sum(da * w) and sum(w over the valid cells) march together inside the
            # reduction kernel -- divided there for one dim, carried side by side through the remaining (small)
            # reductions for several (reference: product, mask, two sum arrays and a quotient array)
            out = self._weighted_reduce(da, factors, dims, "mean_valid" if skip else "mean_all")
        return to_xarray(out) if was_xr else out

    def apply_as_grid_ufunc(self, func, *args, axis=None, signature="", padding_width=None, padding=None,
                            fill_value=None, dask="forbidden", map_overlap=False, pad_before_func=True, **kwargs):
        """Apply a user function on unlabelled arrays in a grid-aware manner (generic path)."""
        return apply_as_grid_ufunc(func, *args, axis=axis, grid=self, signature=signature, padding_width=padding_width,
                                   padding=padding, fill_value=fill_value, dask=dask, map_overlap=map_overlap,
                                   pad_before_func=pad_before_func, **kwargs)

    def vorticity(self, u, v, x_axis: str = "X", y_axis: str = "Y", padding=None, fill_value=None,
                  metric_weighted: bool = True):
        """Fused relative vorticity `(diff(v, X) - diff(u, Y)) / area` in one kernel launch.

        Equivalent (bit for bit) to the chain of three reference operators
        `(grid.diff(v, X) - grid.diff(u, Y)) / grid.get_metric(zeta, (X, Y))` with both diffs
        center->left; the reference's own docs motivate fusing it (docs/grid_ufuncs.md:27).
        On grids with face connections or a north fold `u`, `v` are the X / Y components of a
        vector: the chain is then `(grid.diff({Y: v}, X, other_component={X: u}) -
        grid.diff({X: u}, Y, other_component={Y: v})) / area`, and the two one-cell halos are
        gathered through the topology's token map before the same single launch."""
        (u, xr1), (v, xr2) = self._wrap_in(u), self._wrap_in(v)
        xa, ya = self.axes[x_axis], self.axes[y_axis]
        vx_pos, vx_dim = xa._get_position_name(v)
        uy_pos, uy_dim = ya._get_position_name(u)
        if (vx_pos, uy_pos) != ("center", "center") or "left" not in xa.coords or "left" not in ya.coords:
            raise NotImplementedError("fused vorticity needs v at X:center, u at Y:center and left points on both axes")
        out_x, out_y = xa.coords["left"], ya.coords["left"]
        if u.dims[-2:] != (uy_dim, out_x) or v.dims[-2:] != (out_y, vx_dim) or u.dims[:-2] != v.dims[:-2]:
            raise NotImplementedError("fused vorticity needs u(..., YC, XG) and v(..., YG, XC) with (Y, X) last")
        out_dims = u.dims[:-2] + (out_y, out_x)
        if gridops.is_integer_data(u.data) or gridops.is_integer_data(v.data):
            # integer components: the operator chain itself, each step in numpy's dtype (the fused kernel is float-only)
            kw = dict(padding=padding, fill_value=fill_value)
            res = (self.diff({y_axis: v}, x_axis, other_component={x_axis: u}, **kw)
                   - self.diff({x_axis: u}, y_axis, other_component={y_axis: v}, **kw))
            if metric_weighted:
                res = res / self._resident(self.get_metric(res, (x_axis, y_axis)), res.data)
            return to_xarray(res) if (xr1 or xr2) else res
        area = None
        if metric_weighted:
            area = _aligned_view(self._resident(self.get_metric(_DimsOnly(out_dims), (x_axis, y_axis)), u.data), out_dims)
        bcx, bcy, fx, fy, hx, hy = self._two_component_halos(u, v, x_axis, y_axis, (1, 0), padding, fill_value,
                                                             x_of="v", y_of="u")
        host = not (_is_tensor(u.data) or _is_tensor(v.data))
        out = _dev.vorticity(u.data, v.data, area, bcx, bcy, fx, fy, hx, hy)
        res = DataArray(_dev.tohost(out) if host else out, out_dims)
        res = _reattach_coords([res], self, None, {out_x, out_y}, [u, v])[0]
        return to_xarray(res) if (xr1 or xr2) else res

    def _scalar_halos(self, a, x_axis, y_axis, widths, padding, fill_value):
        """Boundary modes, fill values and (on complex topologies) the pre-gathered one-cell halos of a
        scalar field along both axes, for the fused one-in / two-out operators."""
        bc = self._complete_user_kwargs_using_axis_defaults(padding, "padding")
        fv = self._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")
        modes, halos = {}, {}
        for ax in (x_axis, y_axis):
            if gridops.complex_topology(self, ax):
                halos[ax] = halo_cells(a, self, ax, widths, padding=padding, fill_value=fill_value).data
                modes[ax] = "halo"
            else:
                if bc[ax] is None:
                    raise no_boundary_error(ax)
                halos[ax] = None
                modes[ax] = bc[ax]
        return (modes[x_axis], modes[y_axis], float(fv[x_axis] or 0.0), float(fv[y_axis] or 0.0), halos[x_axis],
                halos[y_axis])

    def _two_component_halos(self, u, v, x_axis, y_axis, widths, padding, fill_value, x_of: str, y_of: str):
        """Boundary modes, fill values and (on complex topologies) the pre-gathered one-cell halos of the
        fused two-component operators: along X the halo of component `x_of`, along Y of `y_of`; `u` is
        the X-component, `v` the Y-component of the vector (rotation / sign rules of `padding.pad`)."""
        bc = self._complete_user_kwargs_using_axis_defaults(padding, "padding")
        fv = self._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")
        comp = {"u": ({x_axis: u}, {y_axis: v}), "v": ({y_axis: v}, {x_axis: u})}
        modes, halos = {}, {}
        for ax, which in ((x_axis, x_of), (y_axis, y_of)):
            if gridops.complex_topology(self, ax):
                arg, other = comp[which]
                halos[ax] = halo_cells(arg, self, ax, widths, padding=padding, fill_value=fill_value,
                                       other_component=dict(other)).data
                modes[ax] = "halo"
            else:
                if bc[ax] is None:
                    raise no_boundary_error(ax)
                halos[ax] = None
                modes[ax] = bc[ax]
        return (modes[x_axis], modes[y_axis], float(fv[x_axis] or 0.0), float(fv[y_axis] or 0.0), halos[x_axis],
                halos[y_axis])

    def divergence(self, u, v, x_axis: str = "X", y_axis: str = "Y", padding=None, fill_value=None,
                   metric_weighted: bool = True):
        """Fused horizontal divergence `(diff(u, X) + diff(v, Y)) / area` in one kernel launch.

        `u` lives on (Y:center, X:left), `v` on (Y:left, X:center); both differences go left -> center
        (the "Divergence" grid ufunc of the reference's docs/ufunc_examples.md, padding_width (0,1)),
        so the result sits at the cell centre and -- with `metric_weighted` -- is divided by the
        (X, Y) metric there.  Bit-identical to the chain of reference operators
        `(grid.diff(u, X) + grid.diff(v, Y)) / grid.get_metric(div, (X, Y))`; pass transports
        (u*dy, v*dx) for the finite-volume form."""
        (u, xr1), (v, xr2) = self._wrap_in(u), self._wrap_in(v)
        xa, ya = self.axes[x_axis], self.axes[y_axis]
        ux_pos, ux_dim = xa._get_position_name(u)
        uy_pos, uy_dim = ya._get_position_name(u)
        vx_pos, vx_dim = xa._get_position_name(v)
        vy_pos, vy_dim = ya._get_position_name(v)
        if (ux_pos, uy_pos, vx_pos, vy_pos) != ("left", "center", "center", "left"):
            raise NotImplementedError("fused divergence needs u at (Y:center, X:left) and v at (Y:left, X:center)")
        out_x, out_y = xa.coords["center"], ya.coords["center"]
        if u.dims[-2:] != (uy_dim, ux_dim) or v.dims[-2:] != (vy_dim, vx_dim) or u.dims[:-2] != v.dims[:-2]:
            raise NotImplementedError("fused divergence needs u(..., YC, XG) and v(..., YG, XC) with (Y, X) last")
        out_dims = u.dims[:-2] + (out_y, out_x)
        if gridops.is_integer_data(u.data) or gridops.is_integer_data(v.data):
            # integer components: the operator chain itself, each step in numpy's dtype (the fused kernel is float-only)
            kw = dict(padding=padding, fill_value=fill_value)
            res = (self.diff({x_axis: u}, x_axis, other_component={y_axis: v}, **kw)
                   + self.diff({y_axis: v}, y_axis, other_component={x_axis: u}, **kw))
            if metric_weighted:
                res = res / self._resident(self.get_metric(res, (x_axis, y_axis)), res.data)
            return to_xarray(res) if (xr1 or xr2) else res
        area = None
        if metric_weighted:
            area = _aligned_view(self._resident(self.get_metric(_DimsOnly(out_dims), (x_axis, y_axis)), u.data), out_dims)
        bcx, bcy, fx, fy, hx, hy = self._two_component_halos(u, v, x_axis, y_axis, (0, 1), padding, fill_value,
                                                             x_of="u", y_of="v")
        host = not (_is_tensor(u.data) or _is_tensor(v.data))
        out = _dev.divergence(u.data, v.data, area, bcx, bcy, fx, fy, hx, hy)
        res = DataArray(_dev.tohost(out) if host else out, out_dims)
        res = _reattach_coords([res], self, None, {out_x, out_y}, [u, v])[0]
        return to_xarray(res) if (xr1 or xr2) else res

    def gradient(self, a, x_axis: str = "X", y_axis: str = "Y", padding=None, fill_value=None,
                 metric_weighted: bool = False):
        """Fused horizontal gradient of a cell-centre field: `(diff(a, X), diff(a, Y))` -- with
        `metric_weighted` `(derivative(a, X), derivative(a, Y))` -- in ONE kernel launch that reads the
        field once (the "Gradient" grid ufunc of the reference's docs/ufunc_examples.md, center -> left,
        padding_width (1, 0) on both axes).  Bit-identical to the two operator calls; on a complex
        topology (face connections / fold) the two one-cell halos of the field are gathered through the
        token map first, then the same single launch runs."""
        a, was_xr = self._wrap_in(a)
        xa, ya = self.axes[x_axis], self.axes[y_axis]
        x_pos, x_dim = xa._get_position_name(a)
        y_pos, y_dim = ya._get_position_name(a)
        if (x_pos, y_pos) != ("center", "center") or "left" not in xa.coords or "left" not in ya.coords:
            raise NotImplementedError("fused gradient needs a field at (Y:center, X:center) and left points on both axes")
        op = self.derivative if metric_weighted else self.diff
        if a.dims[-2:] != (y_dim, x_dim) or gridops.is_integer_data(a.data):  # (integers: the float-only fused kernel is not theirs)
            kw = dict(padding=padding, fill_value=fill_value)
            res = op(a, x_axis, **kw), op(a, y_axis, **kw)
            return tuple(to_xarray(r) for r in res) if was_xr else res
        bcx, bcy, fx, fy, hx, hy = self._scalar_halos(a, x_axis, y_axis, (1, 0), padding, fill_value)
        dims_x = a.dims[:-1] + (xa.coords["left"],)
        dims_y = a.dims[:-2] + (ya.coords["left"], x_dim)
        mx = my = None
        if metric_weighted:
            mx = _aligned_view(self._resident(self.get_metric(_DimsOnly(dims_x), (x_axis,)), a.data), dims_x)
            my = _aligned_view(self._resident(self.get_metric(_DimsOnly(dims_y), (y_axis,)), a.data), dims_y)
        host = not _is_tensor(a.data)
        gx, gy = _dev.gradient(a.data, bcx, bcy, fx, fy, mx, my, hx, hy)
        rx = DataArray(_dev.tohost(gx) if host else gx, dims_x, name=a.name)
        ry = DataArray(_dev.tohost(gy) if host else gy, dims_y, name=a.name)
        rx = _reattach_coords([rx], self, None, {xa.coords["left"]}, [a])[0]
        ry = _reattach_coords([ry], self, None, {ya.coords["left"]}, [a])[0]
        return (to_xarray(rx), to_xarray(ry)) if was_xr else (rx, ry)

    def flux(self, u, v, tracer, x_axis: str = "X", y_axis: str = "Y", padding=None, fill_value=None):
        """Fused first-order advective flux of a cell-centre tracer: `(u * interp(T, X), v * interp(T, Y))`
        in one launch (the "Advection" flux of docs/ufunc_examples.md; u at (Y:center, X:left), v at
        (Y:left, X:center)).  Bit-identical to the chain of reference operators; `padding` /
        `fill_value` pad the tracer (on complex topologies: its pre-gathered halos)."""
        (u, xr1), (v, xr2), (t, xr3) = self._wrap_in(u), self._wrap_in(v), self._wrap_in(tracer)
        xa, ya = self.axes[x_axis], self.axes[y_axis]
        tx_pos, tx_dim = xa._get_position_name(t)
        ty_pos, ty_dim = ya._get_position_name(t)
        if (tx_pos, ty_pos) != ("center", "center") or "left" not in xa.coords or "left" not in ya.coords:
            raise NotImplementedError("fused flux needs the tracer at (Y:center, X:center) and left points on both axes")
        dims_x = t.dims[:-2] + (ty_dim, xa.coords["left"])
        dims_y = t.dims[:-2] + (ya.coords["left"], tx_dim)
        was_xr = xr1 or xr2 or xr3
        if (t.dims[-2:] != (ty_dim, tx_dim) or u.dims != dims_x or v.dims != dims_y
                or any(gridops.is_integer_data(q.data) for q in (u, v, t))):  # integers: each step in numpy's dtype
            kw = dict(padding=padding, fill_value=fill_value)
            res = u * self.interp(t, x_axis, **kw), v * self.interp(t, y_axis, **kw)
            return tuple(to_xarray(r) for r in res) if was_xr else res
        bcx, bcy, fvx, fvy, hx, hy = self._scalar_halos(t, x_axis, y_axis, (1, 0), padding, fill_value)
        host = not (_is_tensor(u.data) or _is_tensor(v.data) or _is_tensor(t.data))
        fx, fy = _dev.flux(u.data, v.data, t.data, bcx, bcy, fvx, fvy, hx, hy)
        rx = DataArray(_dev.tohost(fx) if host else fx, dims_x)
        ry = DataArray(_dev.tohost(fy) if host else fy, dims_y)
        rx = _reattach_coords([rx], self, None, {xa.coords["left"]}, [u, t])[0]
        ry = _reattach_coords([ry], self, None, {ya.coords["left"]}, [v, t])[0]
        return (to_xarray(rx), to_xarray(ry)) if was_xr else (rx, ry)

    def transform(self, da, axis, target, **kwargs):
        """Convert `da` to new 1-D coordinates along `axis` (linear / log / conservative; reference
        grid.py:1687-1777 -> transform.py:284-514), one HIP kernel launch per call."""
        from .transform import transform as _transform

        return _transform(self, axis, da, target, **kwargs)


# ----------------------------------------------------------------------------------------------
def _shifted_dims(grid: Grid, array, ax_name: str, to_pos: str, from_pos: Optional[str] = None) -> Tuple[str, ...]:
    """dims of the result of moving `array` to `to_pos` along `ax_name` (input order kept).  `from_pos`: the position
    the step's signature names -- fixed from the ORIGINAL array before the first axis (xgcm/grid.py:790-794), so a later
    axis is not looked up again on an array a metric product may have given a second dim of that axis"""
    dim = grid.axes[ax_name].coords.get(from_pos) if from_pos is not None else None
    if dim is None or dim not in array.dims:
        _, dim = grid.axes[ax_name]._get_position_name(array)
    new = grid.axes[ax_name].coords.get(to_pos, dim)
    return tuple(new if d == dim else d for d in array.dims)


def _cumsum_trim_pad(pos: str, to: str, reverse: bool, ax) -> Tuple[int, int, int, int]:
    """(trim_lo, trim_hi, pad_lo, pad_hi) of the reference's table (grid.py:1326-1383)."""
    pair = (pos, to)
    natural = (("center", "right"), ("left", "center"))
    shifted = (("center", "left"), ("right", "center"))
    shrink = (("center", "inner"), ("outer", "center"))
    grow = (("center", "outer"), ("inner", "center"))
    if not reverse:
        table = {natural: (0, 0, 0, 0), shifted: (0, 1, 1, 0), shrink: (0, 1, 0, 0), grow: (0, 0, 1, 0)}
    else:
        table = {shifted: (0, 0, 0, 0), natural: (1, 0, 0, 1), shrink: (1, 0, 0, 0), grow: (0, 0, 0, 1)}
    for pairs, widths in table.items():
        if pair in pairs:
            return widths
    raise ValueError(f"From `{pos}` to `{to}` is not a valid position shift for cumsum operation along axis {ax}.")


def _ones_like(data):
    if _is_tensor(data):
        return _dev.synthetic(tuple(data.shape), 0, 0, 0.0, 1.0)
    return np.ones(data.shape)


def _valid_mask(da: DataArray) -> DataArray:
    """1.0 where `da` is not NaN else 0.0, computed on the GPU as (da - da) == 0 -> via min/max-free arithmetic."""
    # x - x is 0 for finite x and NaN for NaN/inf; nan-skipping sum of (x - x + 1) over a length-1 axis gives the mask
    if gridops.is_integer_data(da.data):  # integers and bool hold no NaN (and numpy refuses `bool - bool`)
        return da._replace(data=_ones_like(da.data), coords=OrderedDict(), name=None)
    z = da - da
    one = z + 1.0
    host = not _is_tensor(one.data)
    data = _dev.asdevice(one.data)
    mask = _dev.reduce1d(data[None], 0, None, True)
    return DataArray(_dev.tohost(mask) if host else mask, da.dims)


_SELECT_CACHE: Dict[Tuple[str, Tuple], Any] = {}


def _select_grid_ufunc(funcname, signature: _GridUFuncSignature, module, **kwargs):
    """The one GridUFunc of `module` whose name starts with `funcname` and whose signature is
    equivalent to `signature` (reference grid.py:1779-1824).  Lookups in the built-in `gridops`
    namespace (fixed at import) are memoised: the scan costs ~65 us, a kernel on a small array less."""
    builtin = module is gridops
    key = (funcname, signature._canonical())
    if builtin and key in _SELECT_CACHE:
        return _SELECT_CACHE[key], kwargs
    # (members through getattr, so that a class used as a namespace -- staticmethods -- works like a module)
    named = [f for name, f in inspect.getmembers(module, lambda o: isinstance(o, GridUFunc)) if name.startswith(funcname)]
    if not named:
        raise NotImplementedError(f"Could not find any pre-defined {funcname} grid ufuncs")
    matching = [f for f in named if f.signature.equivalent(signature)]
    if not matching:
        raise NotImplementedError(f"Could not find any pre-defined {funcname} grid ufuncs with signature {signature}")
    if len(matching) > 1:
        raise ValueError(
            f"Function {funcname} with signature='{signature}' and kwargs={kwargs.copy()} is an ambiguous selection"
        )
    if builtin:
        _SELECT_CACHE[key] = matching[0]
    return matching[0], kwargs
