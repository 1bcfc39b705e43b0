"""Boundary padding of labelled arrays on the GPU (generic path for user grid ufuncs).

`pad` has the reference's signature and semantics (xgcm/padding.py:765-871):

* simple topologies -> `_pad_basic` (:575-616): per axis, in `padding_width` order, periodic =
  numpy 'wrap', fill = 'constant', extend = 'edge'; the whole multi-axis pad is ONE kernel
  launch (xg_pad_f64), not one array copy per axis;
* a north fold on a padded axis -> `_pad_fold` (:689-762);
* face connections -> `_pad_face_connections` (:260-572).

The two complex topologies are pure data movement, so their procedure runs once on a plane of
int64 tokens (xgcm_amd/halo_map.py, host-side index logic, cached on the grid) and the data
moves in one `xg_gather_f64` launch.  All coordinates are stripped from the result; all-zero
widths return the input untouched; a padded edge without boundary condition or connection
raises the reference's ValueError.

The built-in diff/interp/min/max/cumsum operators on simple topologies never call this: their
halo is fused into the stencil kernels (xgcm_amd/gridops.py).
"""

from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

import numpy as np

from . import device as _dev
from . import halo_map as _hm
from .labeled import DataArray, _is_tensor, from_xarray, is_xarray, to_xarray

_XGCM_BOUNDARY_KWARG_TO_XARRAY_PAD_KWARG = {"periodic": "wrap", "fill": "constant", "extend": "edge"}


def _np_dtype_of(da):
    data = getattr(da, "data", None)
    return data.dtype if isinstance(data, np.ndarray) else None


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
    def _xr(v):
        return any(_xr(x) for x in v.values()) if isinstance(v, dict) else is_xarray(v)

    if _xr(data) or _xr(other_component):  # called directly with xarray objects: xarray out, as the reference's `pad`
        conv = lambda v: ({k: conv(x) for k, x in v.items()} if isinstance(v, dict) else (from_xarray(v) if is_xarray(v) else v))  # noqa: E731
        out = pad(conv(data), grid, padding_width, padding, fill_value, None if other_component is None else conv(other_component), **kwargs)
        return to_xarray(out) if isinstance(out, DataArray) else out
    halo_only = kwargs.pop("_halo_only", None)  # internal: see `halo_cells`
    dry = kwargs.pop("_dry", False)  # internal (xgcm_amd.lazy): validate and build the halo map, move nothing, return None
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
    data = _strip_all_coords(data)
    connected = getattr(grid, "_connected_axes", ())
    faces = getattr(grid, "_face_connections", None)
    if faces is not None:
        # the reference sends EVERY pad of such a grid through `_pad_face_connections` (xgcm/padding.py:849-857), which
        # walks the faces by number and looks each one up in the connections (:394-396): a face the dict leaves out is
        # its KeyError, whatever is being padded -- raised below, once nothing else has objected
        facedim = grid._facedim
        first = next(iter(data.values())) if isinstance(data, dict) else data
        missing_face = None
        if facedim in first.dims:
            missing_face = next((i for i in range(first.sizes[facedim]) if i not in faces[facedim]), None)
    if faces is not None and (halo_only is not None or any(ax in connected and any(w) for ax, w in padding_width.items())):
        # (padding only axes that no link touches is the ordinary per-axis pad: `_pad_face_connections`
        # pre-pads them with `_pad_basic`, overwrites nothing and trims the other axes back to zero width)
        out = _pad_face_connections(data, grid, padding_width, padding, fill_value, other_component, halo_only, dry)
    elif getattr(grid, "_folds", None) and any(ax in grid._folds for ax in padding_width):
        out = _pad_fold(data, grid, padding_width, padding, fill_value, halo_only, dry)
    else:
        if halo_only is not None:
            raise ValueError("halo-only padding is meant for complex topologies")
        if isinstance(data, dict):
            [data] = list(data.values())
        out = _pad_basic(data, grid, padding_width, padding, fill_value)
    if faces is not None and missing_face is not None:
        raise KeyError(missing_face)  # (after the refusals above: the reference checks boundary conditions before it walks the faces)
    if halo_only is None and isinstance(out, DataArray):
        # `DataArray.pad` keeps the array's attrs, and `numpy.pad` its dtype -- byte order included (the operators go on
        # computing, which makes numpy's results native; a direct call sees the padded array itself)
        first = next(iter(data.values())) if isinstance(data, dict) else data
        out.attrs = dict(getattr(first, "attrs", None) or {})
        want = _np_dtype_of(first)
        if want is not None and not want.isnative and not _is_tensor(out.data) and np.asarray(out.data).dtype != want:
            out = out._replace(data=np.asarray(out.data).astype(want))
    if faces is not None and halo_only is None and isinstance(out, DataArray) and grid._facedim in out.dims \
            and out.dims[0] != grid._facedim:
        # ... and rebuilds the array with `xr.concat(faces, dim=facedim)` (:555): the face dim comes out FIRST.  The
        # operators restore the input's order afterwards (`_restore_input_dim_order`); a direct call sees this one (a view)
        out = out.transpose(grid._facedim, *[d for d in out.dims if d != grid._facedim])
    return out


def _pad_basic(data: DataArray, grid, padding_width, padding, fill_value) -> DataArray:
    """numpy.pad chain over the given axes in one launch (reference padding.py:575-616)."""
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
        if mode not in _XGCM_BOUNDARY_KWARG_TO_XARRAY_PAD_KWARG:
            raise KeyError(mode)
        num = data.get_axis_num(dim)
        widths[num] = (int(w[0]), int(w[1]))
        bc[num] = mode
        f = fill_value[ax]
        fv[num] = 0.0 if f is None else f  # (cast to the array's dtype by the device layer, like numpy.pad)
    if not widths:
        return data
    host = not _is_tensor(data.data)
    out = _dev.pad_nd(data.data, widths, bc, fv)
    return DataArray(_dev.tohost(out) if host else out, data.dims, name=data.name)


# ------------------------------------------------------------------------------------------
# north-fold boundary value `{"fold": pivot[, "south": mode]}`  (semantics: reference padding.py:60-177;
# the texts of the errors are the reference's, pinned by tests/golden/fold_reference.json)
# ------------------------------------------------------------------------------------------
# A tripolar grid's northern boundary is folded about a pole that lies either ON a cell edge or in a cell
# CENTRE, independently along the seam (zonal) and the fold (meridional) axis.  Everything downstream needs
# exactly those two bits, so a pivot is carried as `(seam_on_edge, fold_on_edge)`; the NEMO-style letters
# name the point type the pole coincides with.
_NAMED_POLES = {
    "center": (False, False), "t": (False, False),
    "u": (True, False),
    "v": (False, True),
    "corner": (True, True), "f": (True, True),
}
_SOUTH_MODES = tuple(_XGCM_BOUNDARY_KWARG_TO_XARRAY_PAD_KWARG)  # what the southern edge of the fold axis may do


class FoldSpec(dict):
    """A validated fold padding value.  It stays a plain mapping with the keys `fold` (the pivot as the user
    gave it) and `south` (the mode of the opposite edge) because `Axis.padding` hands it back to user code."""

    @staticmethod
    def looks_like(value) -> bool:
        return isinstance(value, Mapping) and "fold" in value

    @classmethod
    def parse(cls, value) -> "FoldSpec":
        problem = cls._first_problem(value)
        if problem is not None:
            raise ValueError(problem)
        return cls(fold=value["fold"], south=value.get("south", "fill"))

    @staticmethod
    def _first_problem(value) -> Optional[str]:
        """The first rule `value` violates, in the reference's words, or None."""
        if not FoldSpec.looks_like(value):
            return f"Not a fold padding value: {value!r}"
        pivot, south = value["fold"], value.get("south", "fill")
        names = sorted(_NAMED_POLES)
        stray_keys = sorted(k for k in value if k not in ("fold", "south"))
        by_name, by_axis = isinstance(pivot, str), isinstance(pivot, Mapping)
        bad_positions = {ax: pos for ax, pos in pivot.items() if pos not in _hm.HALF_CELL_OFFSET} if by_axis else {}
        rules = (
            (bool(stray_keys),
             lambda: f"Unknown keys {stray_keys} in fold padding {dict(value)!r}. "
                     "Allowed keys are 'fold' (pivot type) and 'south' (south-edge mode)."),
            (not (by_name or by_axis),
             lambda: f"Fold pivot must be a name ({names}) or an {{axis: position}} mapping, got {pivot!r}."),
            (by_name and pivot.lower() not in _NAMED_POLES,
             lambda: f"Unknown fold pivot {pivot!r}. Use one of {names} or an explicit {{axis: position}} mapping."),
            (by_axis and len(pivot) == 0,
             lambda: "Explicit fold pivot mapping must not be empty."),
            (bool(bad_positions),
             lambda: f"Invalid position(s) {bad_positions} in explicit fold pivot {dict(pivot)!r}. "
                     f"Each must be one of {sorted(_hm.HALF_CELL_OFFSET)}."),
            (south not in _SOUTH_MODES,
             lambda: f"Fold 'south' mode must be one of {list(_SOUTH_MODES)}, got {south!r}."),
        )
        return next((text() for violated, text in rules if violated), None)


def pole_on_edges(pivot, fold_axis: str, seam_axis: str) -> Tuple[bool, bool]:
    """`(seam_on_edge, fold_on_edge)` of a validated pivot.  A named pivot is a table look-up; an explicit
    `{axis: position}` pivot puts the pole on a cell edge of every axis it names at a non-centre position,
    unnamed axes keep the pole in the cell centre."""
    if isinstance(pivot, str):
        return _NAMED_POLES[pivot.lower()]
    on_edge = {seam_axis: False, fold_axis: False}
    for axis_name, position in pivot.items():
        if axis_name not in on_edge:
            raise ValueError(
                f"Fold pivot axis {axis_name!r} is neither the fold axis {fold_axis!r} nor the seam axis {seam_axis!r}."
            )
        on_edge[axis_name] = position != "center"
    return on_edge[seam_axis], on_edge[fold_axis]


# ------------------------------------------------------------------------------------------
# token-map padding: shared plumbing
# ------------------------------------------------------------------------------------------
def _axis_dims(grid) -> Dict[str, Tuple[str, ...]]:
    return {name: tuple(ax.coords.values()) for name, ax in grid.axes.items()}


def halo_cells(data, grid, ax_name: str, widths: Tuple[int, int], padding=None, fill_value=None,
               other_component=None) -> DataArray:
    """Only the halo cells `pad` would add along `ax_name` on a complex topology: an array shaped
    like the input with that axis shortened to lo + hi (low halo first).  Feeds
    xg_stencil1d_halo_f64, which then needs no padded copy of the field."""
    return pad(data, grid, {ax_name: tuple(widths)}, padding=padding, fill_value=fill_value,
               other_component=other_component, _halo_only=ax_name)


class InteriorOf(DataArray):
    """The interior of an array that ALREADY holds its padded layout along one dim (cells lo : n - hi), without slicing
    it: it reports the interior's sizes to the builders of a token plane while `.data` stays the whole contiguous buffer,
    and `_gather` reads that buffer through tokens re-indexed to its extents.  `Grid.cumsum` on a connected axis uses it to
    gather the halo cells of the cumulative field from the buffer the scan has just written (reference: pad the trimmed
    cumsum, xgcm/grid.py:1385-1395) -- no padded copy."""

    __slots__ = ("src_pad",)

    def __init__(self, buffer: DataArray, dim: str, lo: int, hi: int):
        super().__init__(buffer.data, buffer.dims, name=buffer.name)
        self.src_pad = (dim, int(lo), int(hi))

    @property
    def shape(self):
        dim, lo, hi = self.src_pad
        return tuple(int(n) - (lo + hi if d == dim else 0) for d, n in zip(self.dims, self.data.shape))

    def _replace(self, data=None, dims=None, coords=None, name="__keep__"):
        if data is not None or dims is not None:
            raise NotImplementedError("the interior view of a padded buffer cannot be re-shaped")
        return self  # (coords are never set on it)


def _reindex_tokens(tokens: np.ndarray, src_sizes, k: int, lo: int, hi: int) -> np.ndarray:
    """tokens counting source cells row-major over `src_sizes` -> the same cells counted over the extents of a buffer
    that carries `lo` / `hi` extra cells along mapped dim number `k` (fill tokens and signs untouched)"""
    t = np.asarray(tokens, dtype=np.int64)
    a = np.abs(t)
    is_cell = (a >= 1) & (a < _hm.FILL_BASE)
    src = np.where(is_cell, a - 1, 0)
    if src.size and int(src.max()) >= int(np.prod(src_sizes)):
        # (no partner component reaches here: the plane was built for faces that are not what the links assume)
        raise ValueError("the face connections swap axes and the faces of this field are no longer square: they cannot "
                         "exchange halos")
    coords = list(np.unravel_index(src, tuple(int(v) for v in src_sizes)))
    coords[k] = coords[k] + int(lo)
    new_sizes = [int(v) + ((lo + hi) if i == k else 0) for i, v in enumerate(src_sizes)]
    moved = np.ravel_multi_index(coords, new_sizes) + 1
    return np.where(is_cell, np.sign(t) * moved, t).astype(np.int64)


def _gather(data: DataArray, partner: Optional[DataArray], grid, key, build, partner_same_as=None,
            halo_dim: Optional[str] = None, dry: bool = False) -> DataArray:
    """Run (or reuse) the token-plane builder `build()` -> (plane, mapped dims, lo per dim, fills)
    and move the data through it.  With `halo_dim` only the halo cells along that dim are produced."""
    cache = grid.__dict__.setdefault("_halo_maps", {})
    src_pad = getattr(data, "src_pad", None)  # `data` is an InteriorOf: its buffer is read in place
    if src_pad is not None and (halo_dim != src_pad[0] or partner is not None):
        raise NotImplementedError("a padded buffer serves as the source of its own halo cells only")
    key = key + (halo_dim, src_pad)
    entry = cache.get(key)
    if entry is None:
        plane, lo_of_dim, fills = build()
        mapped_dims = tuple(d for d in data.dims if d in plane.dims)
        plane = plane.transpose(mapped_dims)
        if halo_dim is not None:
            lo_h = int(lo_of_dim.get(halo_dim, 0))
            n_in = data.sizes[halo_dim]
            plane = _hm.Plane.concat([plane.isel(halo_dim, slice(0, lo_h)), plane.isel(halo_dim, slice(lo_h + n_in, None))],
                                     halo_dim)
            lo_of_dim = dict(lo_of_dim)
            lo_of_dim[halo_dim] = plane.size(halo_dim) + n_in + 1  # no cell of this plane is an interior cell
        tokens_h = np.ascontiguousarray(plane.a)
        if src_pad is not None:
            tokens_h = np.ascontiguousarray(_reindex_tokens(tokens_h, [data.sizes[d] for d in mapped_dims],
                                                            mapped_dims.index(halo_dim), src_pad[1], src_pad[2]))
        entry = {"tokens": tokens_h, "sizes": dict(zip(mapped_dims, plane.a.shape)),
                 "lo": lo_of_dim, "fills": list(fills.values), "device": None}
        if len(cache) > 64:
            cache.clear()
        cache[key] = entry
    if dry:  # the map exists (every check of the topology has run); the caller moves the data later
        return None
    mapped = [d in entry["sizes"] for d in data.dims]
    out_shape = [entry["sizes"].get(d, n) for d, n in zip(data.dims, data.shape)]
    lo = [int(entry["lo"].get(d, 0)) for d in data.dims]
    host = not (_is_tensor(data.data) or (partner is not None and _is_tensor(partner.data)))
    tokens = entry["tokens"]
    if not host:
        if entry["device"] is None:
            entry["device"] = _dev.upload_tokens(tokens)
        tokens = entry["device"]
    perm = None
    pdata = None
    if partner is not None:
        # partner dim k plays the role of out dim perm[k]: unmapped dims by name, mapped dims in order
        free = [i for i, d in enumerate(data.dims) if mapped[i]]
        perm = []
        for d in partner.dims:
            twin = (partner_same_as or {}).get(d, d)  # the padded array's dim on the same grid axis
            if twin in data.dims and not mapped[data.dims.index(twin)]:
                perm.append(data.dims.index(twin))
            else:
                perm.append(free.pop(0))
        pdata = partner.data
    out = _dev.gather(data.data, pdata, tokens, mapped, lo, out_shape, entry["fills"], perm)
    return DataArray(_dev.tohost(out) if host else out, data.dims, name=data.name)


def _fill_key(fill_value: Mapping) -> Tuple:
    return tuple((k, None if v is None else (float(v) if float(v) == float(v) else "nan")) for k, v in fill_value.items())


# ------------------------------------------------------------------------------------------
# north fold (reference padding.py:689-762)
# ------------------------------------------------------------------------------------------
def _pad_fold(data, grid, padding_width, padding, fill_value, halo_only=None, dry: bool = False) -> DataArray:
    isvector = isinstance(data, dict)
    if isvector:
        # a fold is a 180-degree pivot: the lone component flips sign, no partner is needed
        _, data = dict(data).popitem()
    fold_axes = [ax for ax in padding_width if ax in grid._folds and padding_width[ax][1] > 0]
    if len(fold_axes) > 1:
        raise NotImplementedError(
            f"Padding more than one north-fold axis at once is not supported (got fold axes {sorted(fold_axes)})."
        )
    basic_width, basic_padding = {}, {}
    for ax, w in padding_width.items():
        if ax in grid._folds:
            per_call = padding[ax]
            basic_width[ax] = (w[0], 0)
            basic_padding[ax] = per_call if isinstance(per_call, str) else grid._folds[ax]["south"]
        else:
            basic_width[ax] = tuple(w)
            basic_padding[ax] = padding[ax]
    if not fold_axes and halo_only is None:
        return None if dry else _pad_basic(data, grid, basic_width, basic_padding, fill_value)

    fax = fold_axes[0] if fold_axes else halo_only
    info = grid._folds[fax]
    seam_axis = info["seam_axis"]
    pivot = pole_on_edges(info["pivot"], fax, seam_axis)
    fold_position, fold_dim = grid.axes[fax]._get_position_name(data)
    seam_position, seam_dim = grid.axes[seam_axis]._get_position_name(data)
    width = int(padding_width[fax][1])
    dim_of_axis = {fax: fold_dim, seam_axis: seam_dim}
    for ax, w in basic_width.items():
        if any(w):
            dim_of_axis[ax] = grid.axes[ax]._get_position_name(data)[1]
    mapped_dims = tuple(d for d in data.dims if d in dim_of_axis.values())
    sizes = tuple(data.sizes[d] for d in mapped_dims)

    def build():
        fills = _hm.FillTable()
        plane = _hm.identity_plane(sizes, mapped_dims)
        if width > 0:
            plane = _hm.fold_plane(plane, fold_dim, fold_position, seam_dim, seam_position, pivot, width, isvector, fax)
        plane = _hm.basic_pad(plane, dim_of_axis, basic_width, basic_padding, fill_value, fills, no_boundary_error)
        lo = {dim_of_axis[ax]: int(w[0]) for ax, w in basic_width.items() if ax in dim_of_axis}
        return plane, lo, fills

    key = ("fold", data.dims, sizes, fax, tuple((k, tuple(v)) for k, v in padding_width.items()),
           tuple(sorted(basic_padding.items(), key=str)), _fill_key(fill_value), isvector, repr(info["pivot"]))
    return _gather(data, None, grid, key, build, None, None if halo_only is None else dim_of_axis[halo_only], dry)


# ------------------------------------------------------------------------------------------
# face connections (reference padding.py:260-572)
# ------------------------------------------------------------------------------------------
def face_concat_name(grid, da, partner, padding_width):
    """The NAME the reference's face-by-face rebuild leaves on a padded array (xgcm/padding.py:394-555).  Every face is
    put together by `xr.concat([source_slice, target_slice])` for a LEFT link and `[target_slice, source_slice]` for a right
    one, over every connected axis (whether padded or not) with the largest requested width; `concat` names its result after
    its FIRST piece, and the faces are concatenated with face 0 first.  So the name is the one of the source of face 0's
    last left link: the array's own, or -- for a vector component across an axis-swapping link -- its PARTNER's.  The axes
    are walked in the order of the reference's `list(set(...))` (:307-309), rebuilt here the same way; results that depend
    on it are no fixture's business.  (ADVICE r05 asked for a deterministic order instead.  Kept as the reference has it, on
    purpose: the order follows PYTHONHASHSEED there too, and the live differential fuzz -- reference and product in ONE
    process, tests/test_reference_suite_live.py -- compares this very name; the grid's axis order made 17 of 400 calls of
    seed 101 disagree with the reference.  Only the NAME of a padded vector component on an axis-swapping topology is affected.)"""
    links = getattr(grid, "_face_connections", None)
    if links is None or not isinstance(da, DataArray):
        return getattr(da, "name", None)
    facedim = grid._facedim
    named = []
    for c in links[facedim].values():
        named.extend(list(c.keys()))
    pad_axes = list(set(list(set(named)) + list(padding_width.keys())))
    if max([v for w in padding_width.values() for v in w] + [0]) == 0:
        return da.name
    name = da.name
    for ax in pad_axes:
        left = links[facedim].get(0, {}).get(ax, (None, None))[0]
        if left:
            name = partner.name if (partner is not None and left[1] != ax) else da.name
    return name


def _infer_vector_component_axis(grid, da) -> str:
    """Which axis does a bare vector component point along?  A C-grid component sits on cell EDGES along its own
    axis and in cell centres along the others, so the answer is the one axis that contributes a non-centre dim to
    `da` (the rule of the reference's helper of the same name, padding.py:229-257; its error text is kept)."""
    present = set(da.dims)
    edge_axes = [name for name, axis in grid.axes.items()
                 if any(dim in present for position, dim in axis.coords.items() if position != "center")]
    if len(edge_axes) != 1:
        raise ValueError(
            "Could not unambiguously infer the axis of the vector component being "
            f"padded from its staggered position (edge axes found: {edge_axes}). "
            "Pass the component as a `{axis_name: DataArray}` dict so its "
            "orientation is explicit, e.g. "
            "`pad({'Y': v}, ..., other_component={'X': u})`."
        )
    return edge_axes[0]


def _get_all_connection_axes(connections, facedim):
    found = []
    for c in connections[facedim].values():
        for ax in c:
            if ax not in found:
                found.append(ax)
    return found


def _pad_face_connections(da, grid, padding_width, padding, fill_value, other_component=None,
                          halo_only=None, dry: bool = False) -> DataArray:
    facedim, connections = grid._facedim, grid._face_connections
    for what, value in (("Grid connections", connections), ("Face dimension", facedim)):
        if value is None:
            raise ValueError(f"{what} cannot be None")

    vectoraxis = None
    if isinstance(da, dict):
        vectoraxis, da = dict(da).popitem()
    elif other_component is not None:
        vectoraxis = _infer_vector_component_axis(grid, da)
    partner = None
    if vectoraxis is not None:
        if other_component is None:
            raise ValueError("Padding vector components requires `other_component` input.")
        _, partner = dict(other_component).popitem()
        partner = _strip_all_coords(partner)

    # every axis named by a connection or by the request takes part; the reference iterates them
    # in set (hash) order, here: grid axis order (only corner cells can depend on the order)
    wanted = set(_get_all_connection_axes(connections, facedim)) | set(padding_width)
    pad_axes = [ax for ax in grid.axes if ax in wanted] + [ax for ax in padding_width if ax not in grid.axes]
    widths = {ax: tuple(int(v) for v in padding_width.get(ax, (0, 0))) for ax in pad_axes}
    width = max([v for w in widths.values() for v in w] + [0])
    n_face = da.sizes[facedim]
    face_links = connections[facedim]

    # edges without connection need a boundary condition; fully connected axes take a neutral
    # placeholder for the pre-padding that the connection data overwrites (padding.py:338-372)
    prepad_padding = dict(padding)
    for ax in pad_axes:
        if prepad_padding.get(ax) is not None:
            continue
        for side, side_name in ((0, "left"), (1, "right")):
            if widths[ax][side] == 0:
                continue
            loose = [i for i in range(n_face) if face_links.get(i, {}).get(ax, (None, None))[side] is None]
            if loose:
                raise ValueError(
                    f"No boundary condition was specified for axis {ax!r}, "
                    f"but the requested operation needs to pad the {side_name} "
                    f"edge of face(s) {loose}, which have no face "
                    f"connection there. Set a boundary condition, e.g. "
                    f"``padding='fill'`` (or 'extend'/'periodic'), on the Grid "
                    f"(``Grid(..., padding=...)``) or pass ``padding=`` to the "
                    f"grid method."
                )
        prepad_padding[ax] = "fill"
    if width == 0:
        return da

    all_axis_dims = _axis_dims(grid)
    dims_own = {ax: grid.axes[ax]._get_position_name(da)[1] for ax in pad_axes}
    mapped_dims = tuple(d for d in da.dims if d == facedim or d in dims_own.values())
    sizes = tuple(da.sizes[d] for d in mapped_dims)
    p_in = int(np.prod(sizes, dtype=np.int64))
    dims_partner = p_mapped = p_sizes = None
    if partner is not None:
        dims_partner = {ax: grid.axes[ax]._get_position_name(partner)[1] for ax in pad_axes}
        p_mapped = tuple(d for d in partner.dims if d == facedim or d in dims_partner.values())
        p_sizes = tuple(partner.sizes[d] for d in p_mapped)
    uniform = {ax: (width, width) for ax in pad_axes}

    def build():
        fills = _hm.FillTable()
        own = _hm.basic_pad(_hm.identity_plane(sizes, mapped_dims), dims_own, uniform, prepad_padding, fill_value,
                            fills, no_boundary_error)
        other = None
        if partner is not None:
            other = _hm.basic_pad(_hm.identity_plane(p_sizes, p_mapped, offset=p_in), dims_partner, uniform,
                                  prepad_padding, fill_value, fills, no_boundary_error)
        plane = _hm.face_connection_plane(own, other, facedim, face_links, pad_axes, dims_own, dims_partner,
                                          all_axis_dims, widths, width, vectoraxis)
        lo = {dims_own[ax]: widths[ax][0] for ax in pad_axes}
        return plane, lo, fills

    key = ("faces", da.dims, sizes, None if partner is None else (partner.dims, p_sizes), vectoraxis,
           tuple(widths.items()), tuple(sorted(prepad_padding.items(), key=str)), _fill_key(fill_value))
    same_as = None
    if partner is not None:  # unmapped dims of the two components correspond through their grid axis
        same_as = {}
        for d in partner.dims:
            if d in da.dims or d == facedim:
                continue
            for cand in all_axis_dims.values():
                if d in cand:
                    own = [c for c in cand if c in da.dims]
                    if own:
                        same_as[d] = own[0]
    out = _gather(da, partner, grid, key, build, same_as, None if halo_only is None else dims_own[halo_only], dry)
    if isinstance(out, DataArray):
        out = out._replace(name=face_concat_name(grid, da, partner, padding_width))
    return out
