"""Grid keyword arguments read off a dataset's metadata: COMODO coordinate attributes and SGRID topology variables.

The reference builds `coords` for `Grid(ds)` from metadata when `autoparse_metadata` is true, its default
(xgcm/grid.py:151-195 -> xgcm/metadata_parsers.py:4-45): SGRID if the dataset's `Conventions` say so
(xgcm/sgrid.py:6-25), COMODO attributes otherwise (xgcm/comodo.py:23-142).  Host-side attribute parsing, no arithmetic:
restated here so that `Grid(ds)` of an xmitgcm / ROMS dataset works as it does there, with the same error messages.
"""

from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

__all__ = ["parse_metadata", "parse_comodo", "parse_sgrid", "is_sgrid", "assert_valid_sgrid", "get_sgrid_grid",
           "get_all_axes", "get_axis_positions_and_coords"]

_SHIFT_OF = {-0.5: "left", 0.5: "right"}


# ------------------------------------------------------------------------------------------------------
# COMODO: every dimension coordinate names its axis (`axis`) and, unless it holds the cell centres, how it is shifted
# against them (`c_grid_axis_shift` = -0.5 / 0.5); lengths tell inner / outer points from shifted ones
# ------------------------------------------------------------------------------------------------------
def _coordinate_attrs(ds, dim: str) -> Optional[dict]:
    """attrs of the coordinate variable of a dimension (a dimension without one carries no metadata)"""
    try:
        return ds[dim].attrs
    except KeyError:
        return None


def _shift_value(attr):
    """`c_grid_axis_shift` as a number; anything set but not numeric (old xmitgcm wrote such things) counts as 'shifted,
    value unknown' -- True -- so that it is not mistaken for the centre (xgcm/comodo.py:67-78)"""
    if attr is None:
        return None
    try:
        return float(attr)
    except TypeError:
        return True


def _comodo_axes(ds) -> "OrderedDict[str, List[str]]":
    axes: "OrderedDict[str, List[str]]" = OrderedDict()
    for dim in ds.dims:
        attrs = _coordinate_attrs(ds, dim)
        if attrs and "axis" in attrs:
            axes.setdefault(attrs["axis"], []).append(dim)
    return axes


def _comodo_positions(ds, axis: str, dims: List[str]) -> "OrderedDict[str, str]":
    if not dims:
        raise ValueError("Couldn't find any coordinates for axis %s" % axis)
    length = {d: ds.dims[d] for d in dims}
    shift = {d: _shift_value(_coordinate_attrs(ds, d).get("c_grid_axis_shift")) for d in dims}
    unshifted = [d for d in dims if not shift[d]]  # no attribute, or a shift of 0
    if not unshifted:
        raise ValueError("Couldn't find a center coordinate for axis %s" % axis)
    if len(unshifted) > 1:
        raise ValueError("Found two coordinates without `c_grid_axis_shift` attribute for axis %s" % axis)
    center = unshifted[0]
    n = length[center]
    found: "OrderedDict[str, str]" = OrderedDict(center=center)
    for d in dims:
        if d == center:
            continue
        if length[d] == n + 1:
            found["outer"] = d
        elif length[d] == n - 1:
            found["inner"] = d
        elif isinstance(shift[d], float) and shift[d] in _SHIFT_OF:
            side = _SHIFT_OF[shift[d]]
            if length[d] != n:
                raise ValueError("%s coordinate %s has incompatible length %g (axis_len=%g)" % (side.capitalize(), d, length[d], n))
            found[side] = d
        elif shift[d] not in (-0.5, 0.5, 0):
            raise ValueError("Coordinate %s has invalid `c_grid_axis_shift` attribute `%s`. `c_grid_axis_shift` must be one of: "
                             "-0.5, 0.5, 0" % (d, repr(shift[d])))
        else:
            raise ValueError("Coordinate %s has missing `c_grid_axis_shift` attribute `%s`" % (d, repr(shift[d])))
    return found


def _own(ds):
    """an xarray.Dataset read through the product's lazy view (coordinates and attrs; no data variable is loaded)"""
    from .labeled import from_xarray, is_xarray

    return from_xarray(ds) if is_xarray(ds) else ds


def parse_comodo(ds) -> Tuple[object, Dict[str, dict]]:
    """`(ds, {"coords": {axis: {position: dim}}})` from COMODO attributes (the dataset is handed back untouched, the
    return shape of xgcm/metadata_parsers.py:74-97); no attributes, no axes"""
    own = _own(ds)
    return ds, {"coords": {axis: _comodo_positions(own, axis, dims) for axis, dims in _comodo_axes(own).items()}}


# ------------------------------------------------------------------------------------------------------
# SGRID: one variable with cf_role "grid_topology" lists node dimensions and, per cell dimension, the node dimension it
# belongs to and how the cells are padded against the nodes: "xi_rho: xi_psi (padding: both) eta_rho: eta_psi (padding: low)"
# ------------------------------------------------------------------------------------------------------
_PAIR = re.compile(r"(\w+)\s*:\s*(\w+)\s*\(\s*padding\s*:\s*(\w+)\s*\)")
# where the NODES sit relative to cells padded that way (xgcm/sgrid.py:115-121)
_NODE_POSITION = {"high": "left", "low": "right", "both": "inner", "none": "outer"}
_AXIS_INDEX = {"X": 0, "Y": 1, "Z": 2}


def is_sgrid(ds) -> bool:
    for key in ("Conventions", "conventions"):
        if key in ds.attrs:
            return any(tag in ds.attrs[key] for tag in ("SGRID", "sgrid", "Sgrid"))
    return False


def _topology(ds) -> Tuple[str, dict]:
    for name, attrs in ds.variable_attrs().items():  # (attrs only: no variable's data is read for this)
        if attrs.get("cf_role") == "grid_topology":
            return name, attrs
    raise ValueError("Could not find identify SGRID grid in input dataset.")


def _sgrid_axes(name: str, topo: dict) -> List[str]:
    nd = topo["topology_dimension"]
    if nd == 1:
        return ["X"]
    if nd == 2:
        return ["X", "Y", "Z"] if "vertical_dimensions" in topo else ["X", "Y"]
    if nd == 3:
        return ["X", "Y", "Z"]
    raise ValueError(f"SGRID expected dataset with 1-3 spatial dimensions but got {nd} in variable '{name}'.")


def _sgrid_positions(name: str, topo: dict, axis: str) -> "OrderedDict[str, str]":
    if axis not in _AXIS_INDEX:
        raise ValueError(f"Axis name '{axis}' not recognised as one of the default SGRID values 'X', 'Y', 'Z'.")
    nd = topo["topology_dimension"]
    if axis == "Z" and "vertical_dimensions" in topo:  # the vertical of a 2-D topology has an attribute of its own
        pairs = _PAIR.findall(topo["vertical_dimensions"])
        if len(pairs) != 1:
            raise IndexError(f"Found {len(pairs)} vertical_dimensions in grid variable '{name}'. Expecting 1.")
        cell, node, pad = pairs[0]
    else:
        if "node_dimensions" not in topo:
            raise ValueError(f"'node_dimensions' attribute not found in grid variable '{name}''.")
        nodes = topo["node_dimensions"].split()
        i = _AXIS_INDEX[axis]
        if i >= len(nodes):
            raise IndexError(f"Not enough 'node_dimensions'. Expecting {i} got {len(nodes)}.")
        node = nodes[i]
        if nd in (1, 2):
            listing = topo["face_dimensions"]
        elif nd == 3:
            listing = topo["volume_dimensions"]
        else:
            raise ValueError(f"SGRID expected dataset with 1-3 spatial dimensions but got {nd} in variable '{name}'.")
        hits = [p for p in _PAIR.findall(listing) if node in p[1]]
        if len(hits) != 1:
            raise IndexError(f"Found {len(hits)} face_dimensions corresponding to node_dimension '{node}'. Expecting 1.")
        cell, _, pad = hits[0]
    if pad not in _NODE_POSITION:
        raise KeyError(f"Unexpected padding type '{pad}' in SGRID data.")
    return OrderedDict([("center", cell), (_NODE_POSITION[pad], node)])


def parse_sgrid(ds) -> Tuple[object, Dict[str, dict]]:
    name, topo = _topology(_own(ds))
    return ds, {"coords": {axis: _sgrid_positions(name, topo, axis) for axis in _sgrid_axes(name, topo)}}


def parse_metadata(ds) -> Tuple[object, Dict[str, dict]]:
    """`(ds, grid kwargs)` a dataset's metadata provides: SGRID when the conventions attribute says so, COMODO otherwise
    (xgcm/metadata_parsers.py:4-45)"""
    return parse_sgrid(ds) if is_sgrid(ds) else parse_comodo(ds)


# the SGRID helpers under the names of xgcm/sgrid.py (:6, :29, :53, :88)
assert_valid_sgrid = is_sgrid


def get_sgrid_grid(ds) -> str:
    return _topology(_own(ds))[0]


def get_all_axes(ds):
    return dict.fromkeys(_sgrid_axes(*_topology(_own(ds)))).keys()


def get_axis_positions_and_coords(ds, axis_name: str) -> "OrderedDict[str, str]":
    return _sgrid_positions(*_topology(_own(ds)), axis_name)


# ---- the remaining public helpers of the reference's metadata modules --------------------------------------------------
def get_axis_coords(ds, axis_name: str) -> List[str]:
    """dims of `ds` whose coordinate carries `axis == axis_name` (reference `comodo.py:31-52`)"""
    return list(_comodo_axes(_own(ds)).get(axis_name, []))


def assert_valid_comodo(ds) -> None:
    """accepts every dataset, as the reference's does (`comodo.py:11-20`: an unimplemented check)"""


def cf_parser(ds):
    """`(ds, {})`: the reference's CF parser is a placeholder that finds nothing (`metadata_parsers.py:100-119`)"""
    return ds, {}
