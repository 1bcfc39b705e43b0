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
    """A tiny named-dims int64 array: just what the padding procedures need."""

    __slots__ = ("a", "dims")

    def __init__(self, a: np.ndarray, dims: Sequence[str]):
        self.a = a
        self.dims = tuple(dims)
        assert a.ndim == len(self.dims)

    def num(self, dim: str) -> int:
        return self.dims.index(dim)

    def size(self, dim: str) -> int:
        return self.a.shape[self.num(dim)]

    def isel(self, dim: str, index) -> "Plane":
        """slice or integer-array selection along `dim` (the dim is kept)."""
        key = [slice(None)] * self.a.ndim
        key[self.num(dim)] = index
        return Plane(self.a[tuple(key)], self.dims)

    def take_face(self, facedim: str, i: int) -> "Plane":
        n = self.num(facedim)
        return Plane(np.take(self.a, i, axis=n), self.dims[:n] + self.dims[n + 1:])

    def flip(self, dim: str) -> "Plane":
        return Plane(np.flip(self.a, axis=self.num(dim)), self.dims)

    def negate(self) -> "Plane":
        return Plane(-self.a, self.dims)

    def rename(self, mapping: Mapping[str, str]) -> "Plane":
        return Plane(self.a, tuple(mapping.get(d, d) for d in self.dims))

    def swap_names(self, a: str, b: str) -> "Plane":
        """`a` becomes `b`; an existing `b` becomes `a` (higher-dimensional slices)."""
        return Plane(self.a, tuple(b if d == a else (a if d == b else d) for d in self.dims))

    def transpose(self, order: Sequence[str]) -> "Plane":
        return Plane(np.transpose(self.a, [self.num(d) for d in order]), tuple(order))

    def pad(self, dim: str, lo: int, hi: int, mode: str, fill_token: int = 0) -> "Plane":
        if lo == 0 and hi == 0:
            return self
        widths = [(0, 0)] * self.a.ndim
        widths[self.num(dim)] = (lo, hi)
        if mode == "fill":
            return Plane(np.pad(self.a, widths, mode="constant", constant_values=fill_token), self.dims)
        return Plane(np.pad(self.a, widths, mode=_NUMPY_MODE[mode]), self.dims)

    @staticmethod
    def concat(parts: List["Plane"], dim: str) -> "Plane":
        order = parts[0].dims
        arrs = [p.transpose(order).a for p in parts]
        return Plane(np.concatenate(arrs, axis=order.index(dim)), order)

    @staticmethod
    def stack(parts: List["Plane"], dim: str, position: int) -> "Plane":
        order = parts[0].dims
        arrs = [p.transpose(order).a for p in parts]
        return Plane(np.stack(arrs, axis=position), order[:position] + (dim,) + order[position:])


def identity_plane(sizes: Sequence[int], dims: Sequence[str], offset: int = 0) -> Plane:
    n = int(np.prod(sizes, dtype=np.int64)) if len(sizes) else 1
    return Plane((np.arange(n, dtype=np.int64) + (1 + offset)).reshape(tuple(sizes)), dims)


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
# position -> (2 * offset of the point inside its cell, len(dim) - number of cells)
SEAM_POSITION = {"center": (1, 0), "left": (0, 0), "right": (2, 0), "outer": (0, 1), "inner": (2, -1)}


def seam_partner_indices(position: str, pivot_seam: str, length: int) -> np.ndarray:
    """Source column of each seam-axis point mirrored about the pole (padding.py:94-101):
    with the point at `k + offset` cells and the pole on a cell edge (c = 0) or a cell
    centre (c = 1), the partner is `(c - k - 2*offset) mod n_cells`."""
    two_offset, delta = SEAM_POSITION[position]
    n_cells = length - delta
    c = 0 if pivot_seam == "edge" else 1
    return (c - np.arange(length) - two_offset) % n_cells


def fold_plane(plane: Plane, fold_dim: str, fold_position: str, seam_dim: str, seam_position: str,
               pivot: Mapping[str, str], width: int, isvector: bool, fold_axis: str) -> Plane:
    """Append the `width` northern halo rows of a north fold (padding.py:619-686)."""
    fold_kind = "center" if fold_position == "center" else "edge"
    skip = 1 if fold_kind == pivot["fold"] else 0
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
    idx = seam_partner_indices(seam_position, pivot["seam"], plane.size(seam_dim))
    if idx.max() >= plane.size(seam_dim):
        raise NotImplementedError(
            f"A {seam_position!r} seam position is incompatible with a "
            f"center-type fold pivot (seam role {pivot['seam']!r}): the mirror "
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
