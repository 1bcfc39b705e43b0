"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the complex-topology halos.

numpy / pure-Python restatement of the reference's

  * north-fold padding     xgcm/padding.py:82-101 (`_seam_partner_indices`), :619-686
                           (`_fold_north_halo`), :689-762 (`_pad_fold`)
  * face-connection padding xgcm/padding.py:260-572 (`_pad_face_connections`)

working directly on float DATA with explicit index arithmetic (halo bands are overwritten cell
block by cell block on the pre-padded faces).  The product never imports this file: it builds an
int64 token map with another formulation (xgcm_amd/halo_map.py: slice / flip / rename / concat
of a token plane) and moves data on the GPU; the tests compare the two.

Pinning status
--------------
* `seam_partner`, pivot aliases, fold-spec validation: pinned against outputs of the REAL reference
  functions (`_seam_partner_indices`, `_resolve_pivot`, `_parse_fold_padding` import and run under
  oracle/make_golden.py's placeholder modules) -> tests/golden/fold_reference.json.
* fold halos / face-connection halos as a whole: pinned against outputs of the REAL reference functions
  `_pad_face_connections` and `_pad_fold`: oracle/make_golden_topology.py loads the reference's padding.py
  unmodified, gives it a numpy-backed container for the handful of DataArray operations it uses (xarray
  itself is absent) and records 236 seeded cases -- 2-face links of every kind, the cubed sphere, the 13-face
  LLC topology, scalars and vector components, four width sets, three boundary modes; every fold pivot x
  position x width, scalar and vector, including the cases that must raise ->
  tests/golden/topology_reference.{npz,json}, tests/test_topology.py::test_oracle_topology_equals_reference_outputs.
  The explicit known answers of the reference's own tests (xgcm/test/test_fold.py, test_faceconnections.py,
  test_padding.py:341-1205), restated with numpy in tests/test_topology.py, stay as a second pin.
* The reference iterates the padded axes in `set` (hash) order (padding.py:305-307), so where the halos of
  two axes overlap (corners) ITS OWN output depends on PYTHONHASHSEED; here the order is the caller's
  `pad_axes` list, and the golden vectors were recorded under a hash seed that gives the same order (X, Y).

`gather_tokens` is the numpy decode of the product's token map (the kernel's semantics), used as
the checker of `xg_gather_f64` and by the CPU test double oracle/fake_device.py.
"""

from __future__ import annotations

from typing import Mapping, Optional, Sequence, Tuple

import numpy as np

from .refimpl import pad_basic

FILL_BASE = 1 << 62

# offset of a point inside its cell, and how many more points than cells the dim has
_CELL_OFFSET = {"center": 0.5, "left": 0.0, "right": 1.0, "outer": 0.0, "inner": 1.0}
_EXTRA_POINTS = {"center": 0, "left": 0, "right": 0, "outer": 1, "inner": -1}


def seam_partner(position: str, pivot_seam: str, length: int) -> np.ndarray:
    """Column each seam-axis point is mirrored onto (padding.py:82-101), from the physical picture:
    point k sits at x = k + offset (in cells); the pole sits on a cell edge (x = 0) or a cell
    centre (x = 1/2); the mirror image 2*pole - x (mod n_cells) is point number x' - offset."""
    n_cells = length - _EXTRA_POINTS[position]
    off = _CELL_OFFSET[position]
    pole = 0.0 if pivot_seam == "edge" else 0.5
    out = np.empty(length, dtype=np.int64)
    for k in range(length):
        x_mirror = 2.0 * pole - (k + off)
        out[k] = int(round(x_mirror - off)) % n_cells
    return out


def fold_north_halo(a: np.ndarray, fold_num: int, seam_num: int, fold_position: str, seam_position: str,
                    pivot: Mapping[str, str], width: int, isvector: bool) -> np.ndarray:
    """The `width` halo rows above the north edge (padding.py:619-686), ordered upward."""
    n = a.shape[fold_num]
    length = a.shape[seam_num]
    skip = 1 if (("center" if fold_position == "center" else "edge") == pivot["fold"]) else 0
    if width > n - skip:
        raise ValueError("exceeds the interior row(s) available")
    partner = seam_partner(seam_position, pivot["seam"], length)
    if partner.max() >= length:
        raise NotImplementedError("seam position incompatible with a center-type pivot")
    shape = list(a.shape)
    shape[fold_num] = width
    halo = np.empty(shape, dtype=a.dtype)
    for h in range(width):
        src_row = n - 1 - skip - h
        for k in range(length):
            dst = [slice(None)] * a.ndim
            src = [slice(None)] * a.ndim
            dst[fold_num], dst[seam_num] = h, k
            src[fold_num], src[seam_num] = src_row, int(partner[k])
            halo[tuple(dst)] = -a[tuple(src)] if isvector else a[tuple(src)]
    return halo


def pad_fold(a: np.ndarray, axis_num: Mapping[str, int], positions: Mapping[str, str], fold_axis: str,
             seam_axis: str, pivot: Mapping[str, str], south: str, padding_width: Mapping[str, Tuple[int, int]],
             padding: Mapping[str, Optional[str]], fill_value: Mapping[str, float], isvector: bool = False):
    """`_pad_fold` (padding.py:689-762): fold halo from the unpadded interior first, then the
    ordinary per-axis pads (fold axis: south edge only, per-call string mode or `south`)."""
    out = a
    width = padding_width.get(fold_axis, (0, 0))[1]
    if width > 0:
        halo = fold_north_halo(a, axis_num[fold_axis], axis_num[seam_axis], positions[fold_axis],
                               positions[seam_axis], pivot, width, isvector)
        out = np.concatenate([a, halo], axis=axis_num[fold_axis])
    for ax, w in padding_width.items():
        if ax == fold_axis:
            mode = padding.get(ax) if isinstance(padding.get(ax), str) else south
            out = pad_basic(out, axis_num[ax], (w[0], 0), mode, fill_value.get(ax, 0.0))
        else:
            out = pad_basic(out, axis_num[ax], tuple(w), padding.get(ax), fill_value.get(ax, 0.0))
    return out


# ------------------------------------------------------------------------------------------
# face connections
# ------------------------------------------------------------------------------------------
def pad_face_connections(a: np.ndarray, dims: Sequence[str], facedim: str, axis_dim: Mapping[str, str],
                         links: Mapping, pad_axes: Sequence[str], padding_width: Mapping[str, Tuple[int, int]],
                         padding: Mapping[str, Optional[str]], fill_value: Mapping[str, float],
                         partner: Optional[np.ndarray] = None, partner_dims: Optional[Sequence[str]] = None,
                         partner_axis_dim: Optional[Mapping[str, str]] = None,
                         vectoraxis: Optional[str] = None) -> np.ndarray:
    """`_pad_face_connections` (padding.py:260-572) on plain arrays.

    `dims` names the dims of `a`; `axis_dim[ax]` is the dim of `a` on grid axis `ax`.  For a
    vector component `vectoraxis` names its axis and `partner` is the other component (its own
    `partner_dims` / `partner_axis_dim`)."""
    dims = tuple(dims)
    widths = {ax: tuple(padding_width.get(ax, (0, 0))) for ax in pad_axes}
    W = max([v for w in widths.values() for v in w] + [0])
    if W == 0:
        return a
    modes = {ax: (padding.get(ax) if padding.get(ax) is not None else "fill") for ax in pad_axes}

    def prepad(arr, arr_dims, ax_dim):
        out = arr
        for ax in pad_axes:
            fv = fill_value.get(ax)
            out = pad_basic(out, tuple(arr_dims).index(ax_dim[ax]), (W, W), modes[ax], 0.0 if fv is None else fv)
        return out

    own = prepad(a, dims, axis_dim)
    other = other_dims = None
    if vectoraxis is not None:
        if partner is None:
            raise ValueError("Padding vector components requires `other_component` input.")
        other = prepad(partner, partner_dims, partner_axis_dim)
        # the partner seen with the target's dim names: its dim on axis ax is called axis_dim[ax]
        rename = {partner_axis_dim[ax]: axis_dim[ax] for ax in pad_axes}
        other_dims = tuple(rename.get(d, d) for d in partner_dims)

    fnum = dims.index(facedim)
    result = own.copy()
    for f in range(a.shape[fnum]):
        for ax in pad_axes:
            left, right = links.get(f, {}).get(ax, (None, None))
            for link, is_right in ((left, False), (right, True)):
                if not link:
                    continue
                src_face, src_axis, reverse = link
                swap = src_axis != ax
                from_partner = vectoraxis is not None and swap
                S, S_dims = (other, other_dims) if from_partner else (own, dims)
                o_dim = axis_dim[ax]            # direction normal to the edge, in the target
                s_dim = axis_dim[src_axis]      # the source's normal direction (target naming)
                Ls = S.shape[S_dims.index(s_dim)]
                Lo = own.shape[dims.index(o_dim)]
                sign = 1.0
                if vectoraxis is not None:
                    if reverse and vectoraxis == ax:
                        sign = -sign
                    if swap and not reverse and vectoraxis != ax:
                        sign = -sign
                for j in range(W):
                    if is_right:
                        s_idx = (Ls - W - 1 - j) if reverse else (W + j)
                        o_idx = Lo - W + j
                    else:
                        s_idx = (2 * W - 1 - j) if reverse else (Ls - 2 * W + j)
                        o_idx = j
                    # source slab: one index along its normal dim, everything else kept
                    src = [slice(None)] * S.ndim
                    src[S_dims.index(facedim)] = src_face
                    src[S_dims.index(s_dim)] = s_idx
                    slab = S[tuple(src)]
                    slab_dims = [d for d in S_dims if d not in (facedim, s_dim)]
                    if swap:
                        # the source's dim called like the target's normal dim runs along the edge
                        # and lands on the target's dim called like the source's normal dim
                        k = slab_dims.index(o_dim)
                        if not reverse:
                            slab = np.flip(slab, axis=k)
                        slab_dims[k] = s_dim
                    dst = [slice(None)] * own.ndim
                    dst[fnum] = f
                    dst[dims.index(o_dim)] = o_idx
                    dst_dims = [d for d in dims if d not in (facedim, o_dim)]
                    slab = np.transpose(slab, [slab_dims.index(d) for d in dst_dims])
                    result[tuple(dst)] = sign * slab if sign != 1.0 else slab
    # trim the uniform halo back to the requested widths
    for ax in pad_axes:
        lo, hi = widths[ax]
        n = dims.index(axis_dim[ax])
        sel = [slice(None)] * result.ndim
        sel[n] = slice(W - lo, result.shape[n] - (W - hi))
        result = result[tuple(sel)]
    return result


# ------------------------------------------------------------------------------------------
# numpy decode of the product's token map (semantics of xg_gather_f64, include/xgcm_hip.h)
# ------------------------------------------------------------------------------------------
def gather_tokens(x: np.ndarray, partner: Optional[np.ndarray], tokens: np.ndarray, mapped: Sequence[bool],
                  lo: Sequence[int], out_shape: Sequence[int], fills: Sequence[float],
                  partner_perm: Optional[Sequence[int]] = None) -> np.ndarray:
    x = np.asarray(x)
    nd = x.ndim
    mapped = [bool(m) for m in mapped]
    m_dims = [d for d in range(nd) if mapped[d]]
    u_dims = [d for d in range(nd) if not mapped[d]]
    # bring unmapped dims first, mapped dims last (both in order), flatten the mapped block
    if int(np.prod([int(v) for v in out_shape])) == 0:
        return np.empty([int(v) for v in out_shape], dtype=x.dtype)
    xs = np.transpose(x, u_dims + m_dims).reshape([x.shape[d] for d in u_dims] + [int(np.prod([x.shape[d] for d in m_dims]))])
    p_in = xs.shape[-1]
    sources = xs
    if partner is not None:
        partner = np.asarray(partner)
        perm = list(partner_perm) if partner_perm is not None else list(range(nd))
        pm = [k for k in range(nd) if mapped[perm[k]]]                 # partner's mapped dims, own order
        pu = sorted((k for k in range(nd) if not mapped[perm[k]]), key=lambda k: perm[k])  # in out order
        ps = np.transpose(partner, pu + pm).reshape([partner.shape[k] for k in pu] + [int(np.prod([partner.shape[k] for k in pm]))])
        sources = np.concatenate([xs, ps], axis=-1)
    t = np.asarray(tokens, dtype=np.int64).reshape(-1)
    a = np.abs(t)
    is_fill = a >= FILL_BASE
    k = np.where(is_fill, 0, a - 1)
    vals = sources[..., k]
    if is_fill.any():
        fv = np.asarray(list(fills) + [0.0], dtype=x.dtype)
        slot = np.where(is_fill, a - FILL_BASE, 0)
        vals = np.where(is_fill, fv[np.minimum(slot, len(fv) - 1)], vals)
    vals = np.where(t < 0, -vals, vals)
    out_m = [out_shape[d] for d in m_dims]
    vals = vals.reshape([x.shape[d] for d in u_dims] + out_m)
    inv = np.argsort(u_dims + m_dims)
    out = np.transpose(vals, inv)
    # interior cells copy the input (the kernel does not consult the map there)
    out = np.ascontiguousarray(out)
    sel_out, sel_in = [], []
    for d in range(nd):
        if not mapped[d]:
            sel_out.append(slice(None)); sel_in.append(slice(None))
            continue
        a0, a1 = max(int(lo[d]), 0), min(int(lo[d]) + x.shape[d], int(out_shape[d]))
        if a1 <= a0:
            sel_out = None  # a halo-only plane: no interior cell at all
            break
        sel_out.append(slice(a0, a1)); sel_in.append(slice(a0 - int(lo[d]), a1 - int(lo[d])))
    if sel_out is not None:
        out[tuple(sel_out)] = x[tuple(sel_in)]
    assert list(out.shape) == [int(v) for v in out_shape]
    return out
