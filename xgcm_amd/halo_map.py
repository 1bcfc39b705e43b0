"""Halo source maps for complex topologies: north fold and face connections.

The reference pads such grids by slicing, flipping, renaming and concatenating xarray objects
(`_pad_face_connections` xgcm/padding.py:260-572, `_fold_north_halo` / `_pad_fold` :619-762).
Every one of those steps only MOVES values (optionally negating them), so the whole procedure
is a fixed gather: padded cell -> (source array, source cell, sign) or a fill value.  This
module runs the procedure once on a small plane of int64 *tokens* instead of on data -- host
side index arithmetic over the touched ("mapped") dims only -- and the data of every other dim
(time, depth, ...) is then moved on the GPU in one pass by `xg_gather_f64` through that map.

Token encoding (mirrors include/xgcm_hip.h):
    +-(1 + k)           element k (row-major over the mapped dims) of the padded array itself,
                        or of the other vector component when k >= P_in (then k - P_in)
    +-(2**62 + slot)    fill value number `slot`
    negative            the value is negated
"""

from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

FILL_BASE = 1 << 62

_NUMPY_MODE = {"periodic": "wrap", "fill": "constant", "extend": "edge"}


class Plane:
    """A named-dims int64 token array that is evaluated LAZILY.

    The padding procedures below build the padded plane by slicing, flipping, renaming and concatenating
    (as the reference does with data).  Doing that on materialised arrays costs O(plane) memory traffic per
    step -- 2 GB and seconds for the 13 x 4320 x 4320 LLC grid -- although a built-in operator only needs the
    one-cell halo slab of the result.  So every step returns a small recipe instead: `eval(sel)` produces the
    tokens of an outer-product selection (one index array per dim) by pushing the selection down the recipe,
    and the cost of a halo map is proportional to the halo.  `.a` evaluates everything (full pads)."""

    __slots__ = ("dims", "shape", "_full")

    def __init__(self, dims: Sequence[str], shape: Sequence[int]):
        self.dims = tuple(dims)
        self.shape = tuple(int(n) for n in shape)
        self._full = None
        assert len(self.dims) == len(self.shape)

    # -- evaluation ------------------------------------------------------------------------
    def eval(self, sel: Sequence[np.ndarray]) -> np.ndarray:  # pragma: no cover - abstract
        raise NotImplementedError

    @property
    def a(self) -> np.ndarray:
        if self._full is None:
            self._full = self.eval([np.arange(n, dtype=np.int64) for n in self.shape])
        return self._full

    # -- the operations the padding procedures use -----------------------------------------
    def num(self, dim: str) -> int:
        return self.dims.index(dim)

    def size(self, dim: str) -> int:
        return self.shape[self.num(dim)]

    def _identity_maps(self):
        return [(k, np.arange(n, dtype=np.int64), None, 0) for k, n in enumerate(self.shape)]

    def _view(self, dims, maps, fixed=None, negate=False) -> "Plane":
        return _View(self, dims, maps, fixed or {}, -1 if negate else 1)

    def isel(self, dim: str, index) -> "Plane":
        """slice or integer-array selection along `dim` (the dim is kept)."""
        k = self.num(dim)
        maps = self._identity_maps()
        maps[k] = (k, np.arange(self.shape[k], dtype=np.int64)[index], None, 0)
        return self._view(self.dims, maps)

    def take_face(self, facedim: str, i: int) -> "Plane":
        n = self.num(facedim)
        maps = [m for k, m in enumerate(self._identity_maps()) if k != n]
        return self._view(self.dims[:n] + self.dims[n + 1:], maps, fixed={n: int(i)})

    def flip(self, dim: str) -> "Plane":
        return self.isel(dim, slice(None, None, -1))

    def negate(self) -> "Plane":
        return self._view(self.dims, self._identity_maps(), negate=True)

    def rename(self, mapping: Mapping[str, str]) -> "Plane":
        return self._view(tuple(mapping.get(d, d) for d in self.dims), self._identity_maps())

    def swap_names(self, a: str, b: str) -> "Plane":
        """`a` becomes `b`; an existing `b` becomes `a` (higher-dimensional slices)."""
        return self._view(tuple(b if d == a else (a if d == b else d) for d in self.dims), self._identity_maps())

    def transpose(self, order: Sequence[str]) -> "Plane":
        order = tuple(order)
        if order == self.dims:
            return self
        ident = self._identity_maps()
        return self._view(order, [ident[self.num(d)] for d in order])

    def pad(self, dim: str, lo: int, hi: int, mode: str, fill_token: int = 0) -> "Plane":
        """numpy.pad along one dim: wrap / edge / constant (`fill_token`)."""
        if lo == 0 and hi == 0:
            return self
        k = self.num(dim)
        n = self.shape[k]
        idx = np.pad(np.arange(n, dtype=np.int64), (lo, hi), mode="constant" if mode == "fill" else _NUMPY_MODE[mode])
        fill = None
        if mode == "fill":
            fill = np.zeros(n + lo + hi, dtype=np.int64)
            fill[:lo] = fill_token
            fill[n + lo:] = fill_token
        maps = self._identity_maps()
        maps[k] = (k, idx, fill, 1)
        return self._view(self.dims, maps)

    @staticmethod
    def concat(parts: List["Plane"], dim: str) -> "Plane":
        order = parts[0].dims
        return _Concat([p.transpose(order) for p in parts], order.index(dim))

    @staticmethod
    def stack(parts: List["Plane"], dim: str, position: int) -> "Plane":
        order = parts[0].dims
        return _Stack([p.transpose(order) for p in parts], dim, position)


class _Base(Plane):
    """tokens 1 + offset + row-major index: the array itself (or, with an offset, the other component)."""

    __slots__ = ("offset", "strides")

    def __init__(self, sizes, dims, offset=0):
        super().__init__(dims, sizes)
        self.offset = int(offset)
        st, acc = [], 1
        for n in reversed(self.shape):
            st.append(acc)
            acc *= n
        self.strides = tuple(reversed(st))

    def eval(self, sel):
        out = np.full((1,) * len(self.shape), 1 + self.offset, dtype=np.int64)
        for k, s in enumerate(sel):
            shp = [1] * len(self.shape)
            shp[k] = len(s)
            out = out + (np.asarray(s, dtype=np.int64) * self.strides[k]).reshape(shp)
        return out


class _View(Plane):
    """Per-dim index maps (+ fill overrides, + sign) on top of another plane.  `maps[k]` describes dim k of
    the view: (inner dim, index array into it, fill tokens or None, order in which the fill was applied);
    `fixed` pins the remaining inner dims to one index.  Views of views are composed at construction."""

    __slots__ = ("inner", "maps", "fixed", "sign")

    def __init__(self, inner: Plane, dims, maps, fixed, sign):
        if isinstance(inner, _View):  # compose: at most one view layer above any concat / stack / base
            base_prio = max([m[3] for m in inner.maps] + [0])
            by_inner_dim = {}
            for k_in, m in enumerate(inner.maps):
                by_inner_dim[k_in] = m
            new_maps = []
            for (k_in, idx, fill, prio) in maps:
                pos, idx_in, fill_in, prio_in = by_inner_dim[k_in]
                safe = np.where(fill != 0, 0, idx) if fill is not None else idx
                nidx = idx_in[safe]
                nfill, nprio = None, prio_in
                if fill_in is not None:
                    nfill = fill_in[safe] * sign
                if fill is not None:
                    nfill = fill.copy() if nfill is None else np.where(fill != 0, fill, nfill)
                    nprio = base_prio + prio
                new_maps.append((pos, nidx, nfill, nprio))
            new_fixed = dict(inner.fixed)
            for k_in, v in fixed.items():  # an inner VIEW dim pinned: pin what it points at
                pos, idx_in, fill_in, _ = by_inner_dim[k_in]
                if fill_in is not None and fill_in[v] != 0:
                    raise NotImplementedError("pinning a dim at a fill cell")
                new_fixed[pos] = int(idx_in[v])
            maps, fixed, sign, inner = new_maps, new_fixed, sign * inner.sign, inner.inner
        super().__init__(dims, [len(m[1]) for m in maps])
        self.inner, self.maps, self.fixed, self.sign = inner, list(maps), dict(fixed), int(sign)

    def eval(self, sel):
        nd_in = len(self.inner.shape)
        inner_sel = [None] * nd_in
        for (pos, idx, fill, _), s in zip(self.maps, sel):
            j = idx[s]
            if fill is not None:
                j = np.where(fill[s] != 0, 0, j)
            inner_sel[pos] = j
        for pos, v in self.fixed.items():
            inner_sel[pos] = np.array([v], dtype=np.int64)
        res = self.inner.eval(inner_sel)
        order = [m[0] for m in self.maps] + sorted(self.fixed)
        res = np.transpose(res, order).reshape([len(s) for s in sel])
        if self.sign < 0:
            res = -res
        elif not res.flags.writeable or any(m[2] is not None for m in self.maps):
            res = np.array(res)
        for k in sorted(range(len(self.maps)), key=lambda k: self.maps[k][3]):  # later fills win
            fill = self.maps[k][2]
            if fill is None:
                continue
            f = fill[sel[k]]
            hit = np.nonzero(f)[0]
            if hit.size:
                key = [slice(None)] * res.ndim
                key[k] = hit
                shp = [1] * res.ndim
                shp[k] = hit.size
                res[tuple(key)] = f[hit].reshape(shp)
        return res


class _Concat(Plane):
    __slots__ = ("parts", "axis", "bounds")

    def __init__(self, parts: List[Plane], axis: int):
        shape = list(parts[0].shape)
        shape[axis] = sum(p.shape[axis] for p in parts)
        super().__init__(parts[0].dims, shape)
        self.parts, self.axis = list(parts), int(axis)
        self.bounds = np.cumsum([0] + [p.shape[axis] for p in parts])

    def eval(self, sel):
        s = np.asarray(sel[self.axis], dtype=np.int64)
        out = np.empty([len(x) for x in sel], dtype=np.int64)
        for p, b0, b1 in zip(self.parts, self.bounds[:-1], self.bounds[1:]):
            hit = np.nonzero((s >= b0) & (s < b1))[0]
            if not hit.size:
                continue
            sub = list(sel)
            sub[self.axis] = s[hit] - b0
            key = [slice(None)] * out.ndim
            key[self.axis] = hit
            out[tuple(key)] = p.eval(sub)
        return out


class _Stack(Plane):
    __slots__ = ("parts", "position")

    def __init__(self, parts: List[Plane], dim: str, position: int):
        order = parts[0].dims
        super().__init__(order[:position] + (dim,) + order[position:],
                         parts[0].shape[:position] + (len(parts),) + parts[0].shape[position:])
        self.parts, self.position = list(parts), int(position)

    def eval(self, sel):
        faces = np.asarray(sel[self.position], dtype=np.int64)
        sub = [x for k, x in enumerate(sel) if k != self.position]
        out = np.empty([len(x) for x in sel], dtype=np.int64)
        cache = {}
        for j, f in enumerate(faces):
            f = int(f)
            if f not in cache:
                cache[f] = self.parts[f].eval(sub)
            key = [slice(None)] * out.ndim
            key[self.position] = j
            out[tuple(key)] = cache[f]
        return out


def identity_plane(sizes: Sequence[int], dims: Sequence[str], offset: int = 0) -> Plane:
    return _Base(tuple(int(n) for n in sizes), dims, offset)


class FillTable:
    """Fill values referenced by tokens; one slot per distinct value (NaN == NaN here)."""

    def __init__(self):
        self.values: List[float] = []

    def token(self, value) -> int:
        v = 0.0 if value is None else float(value)
        for i, old in enumerate(self.values):
            if old == v or (old != old and v != v):
                return FILL_BASE + i
        if len(self.values) >= 8:
            raise NotImplementedError("more than 8 distinct fill values in one padding call")
        self.values.append(v)
        return FILL_BASE + len(self.values) - 1


def basic_pad(plane: Plane, dim_of_axis: Mapping[str, str], widths: Mapping[str, Tuple[int, int]],
              modes: Mapping[str, Optional[str]], fill_values: Mapping[str, float], fills: FillTable,
              no_boundary_error) -> Plane:
    """The reference's `_pad_basic` (padding.py:575-616) on a token plane: axis by axis in
    `widths` order, numpy.pad wrap / constant / edge; zero-width axes are skipped."""
    for ax, (lo, hi) in widths.items():
        if lo == 0 and hi == 0:
            continue
        mode = modes[ax]
        if mode is None:
            raise no_boundary_error(ax)
        if mode not in _NUMPY_MODE:
            raise KeyError(mode)
        plane = plane.pad(dim_of_axis[ax], int(lo), int(hi), mode, fills.token(fill_values.get(ax)) if mode == "fill" else 0)
    return plane


# ------------------------------------------------------------------------------------------
# north fold
# ------------------------------------------------------------------------------------------
# Points of a staggered axis in HALF-cell units: point k of a dim sits at 2*k + HALF_CELL_OFFSET half cells from
# the western wall of cell 0; a dim of len(dim) points spans len(dim) - EXTRA_POINTS cells.
HALF_CELL_OFFSET = {"center": 1, "left": 0, "right": 2, "outer": 0, "inner": 2}
EXTRA_POINTS = {"center": 0, "left": 0, "right": 0, "outer": 1, "inner": -1}  # len(dim) - number of cells


def mirror_columns(position: str, pole_on_edge: bool, length: int) -> np.ndarray:
    """For every point of the seam dim, the index of the point the fold maps onto it.

    Crossing the pole reflects the zonal coordinate: in half-cell units a point at h lands on 2*p - h, where the
    pole sits at p = 0 (on the wall of cell 0) or p = 1 (in the centre of cell 0).  The image is again a point of
    the same position when 2*p - h differs from the offset by a whole number of cells; wrapped around the
    periodic seam that is index ((2*p - h) - offset) / 2 modulo the number of cells.  (Same map as the
    reference's `_seam_partner_indices`, padding.py:94-101; outputs pinned by tests/golden/fold_reference.json.)"""
    offset = HALF_CELL_OFFSET[position]
    cells = length - EXTRA_POINTS[position]
    here = 2 * np.arange(length) + offset
    image = (0 if pole_on_edge else 2) - here
    return ((image - offset) // 2) % cells


def fold_plane(plane: Plane, fold_dim: str, fold_position: str, seam_dim: str, seam_position: str,
               pivot: Tuple[bool, bool], width: int, isvector: bool, fold_axis: str) -> Plane:
    """Append the `width` northern halo rows of a north fold (padding.py:619-686)."""
    seam_on_edge, fold_on_edge = pivot
    # rows ON the pole line are their own images: present iff the field's fold-axis points share the pole's type
    skip = 1 if (fold_position != "center") == fold_on_edge else 0
    n = plane.size(fold_dim)
    n_interior = n - skip
    if width > n_interior:
        raise ValueError(
            f"North-fold halo width {width} requested on fold axis "
            f"{fold_axis!r} exceeds the {n_interior} interior row(s) available "
            f"to mirror along {fold_dim!r} (grid length {n}"
            f"{f', minus {skip} redundant pole row' if skip else ''}). "
            "The fold can supply at most that many halo rows."
        )
    rows = np.arange(n - 1 - skip, n - 1 - skip - width, -1)  # north to south
    halo = plane.isel(fold_dim, rows)
    idx = mirror_columns(seam_position, seam_on_edge, plane.size(seam_dim))
    if idx.max() >= plane.size(seam_dim):
        raise NotImplementedError(
            f"A {seam_position!r} seam position is incompatible with a "
            f"center-type fold pivot (seam role {'edge' if seam_on_edge else 'center'!r}): the mirror "
            "about a cell-center pole has no partner on this sublattice. Use an "
            "edge-type pivot, or a center/left/right/outer seam position."
        )
    halo = halo.isel(seam_dim, idx)
    if isvector:
        halo = halo.negate()
    return Plane.concat([plane, halo], fold_dim)


# ------------------------------------------------------------------------------------------
# face connections
# ------------------------------------------------------------------------------------------
def face_connection_plane(own: Plane, partner: Optional[Plane], facedim: str, links: Mapping,
                          pad_axes: Sequence[str], dims_own: Mapping[str, str],
                          dims_partner: Optional[Mapping[str, str]], all_axis_dims: Mapping[str, Sequence[str]],
                          widths: Dict[str, Tuple[int, int]], width: int,
                          vectoraxis: Optional[str]) -> Plane:
    """Replace the halos of the connected edges (padding.py:391-547).

    `own` / `partner` are the planes of the component being padded and of the other component,
    both already basic-padded by `width` on every axis of `pad_axes` (the reference's
    "prepadded" arrays); sources are always taken from THOSE, never from faces that already
    received connection data.  `dims_own[ax]` is the dim of `own` on axis `ax`."""
    isvector = vectoraxis is not None
    n_face = own.size(facedim)
    faces: List[Plane] = []
    for i in range(n_face):
        target = own.take_face(facedim, i)
        face_links = links.get(i, {})
        for axname in pad_axes:
            left, right = face_links.get(axname, (None, None))
            target_dim = dims_own[axname]
            for connection, is_right in ((left, False), (right, True)):
                if width == 0 or not connection:
                    continue
                source_face, source_axis, reverse = connection
                swap = axname != source_axis
                if isvector and swap:
                    source = partner.take_face(facedim, source_face)
                    # give the partner's dims the names of the target's positions
                    rename = {}
                    for d in target.dims:
                        if d in source.dims:
                            continue
                        for cand in all_axis_dims.values():
                            if d in cand:
                                src_d = [c for c in cand if c in source.dims][0]
                                rename[src_d] = d
                    source = source.rename(rename)
                else:
                    source = own.take_face(facedim, source_face)
                source_dim = [d for d in all_axis_dims[source_axis] if d in source.dims][0]
                if is_right:
                    src_sel = slice(-2 * width, -width) if reverse else slice(width, 2 * width)
                    tgt_sel = slice(0, -width)
                else:
                    src_sel = slice(width, 2 * width) if reverse else slice(-2 * width, -width)
                    tgt_sel = slice(width, None)
                piece = source.isel(source_dim, src_sel)
                kept = target.isel(target_dim, tgt_sel)
                if swap:
                    piece = piece.swap_names(source_dim, target_dim) if target_dim in piece.dims else piece.rename(
                        {source_dim: target_dim})
                ortho, tangential = target_dim, source_dim
                if reverse:
                    piece = piece.flip(ortho)
                    if isvector and vectoraxis == axname:
                        piece = piece.negate()
                if swap and not reverse:
                    piece = piece.flip(tangential)
                    if isvector and vectoraxis != axname:
                        piece = piece.negate()
                piece = piece.transpose(kept.dims)
                target = Plane.concat([kept, piece] if is_right else [piece, kept], target_dim)
        faces.append(target)
    out = Plane.stack(faces, facedim, own.num(facedim))
    # trim the uniform `width` halos back to the requested widths (padding.py:552-572)
    for axname, (lo, hi) in widths.items():
        dim = dims_own[axname]
        start = width - lo
        stop = width - hi
        out = out.isel(dim, slice(start, -stop if stop else None))
    return out
