"""`Axis`: one direction of a staggered grid and the dims that sit at each cell position.

Host-side metadata only (no arithmetic).  Behaviour and error messages follow reference
xgcm/axis.py:20-257: position validation (:99-123), default shifts / FALLBACK_SHIFTS
(:11-17,126-146), padding / fill_value defaults (:152-171), `_get_position_name` (:232-251),
`_get_axis_dim_num` (:253-256).  A north-fold padding spec (dict value, :153-156) is validated
here and resolved against the seam axis by `Grid._validate_folds`.
"""

from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple, Union

POSITIONS = ("center", "left", "right", "inner", "outer")
VALID_POSITION_NAMES = "|".join(POSITIONS)
VALID_PADDINGS = ("periodic", "fill", "extend")

# where `to` defaults to when the caller does not say (reference axis.py:11-17)
FALLBACK_SHIFTS = {
    "center": ("left", "right", "outer", "inner"),
    "left": ("center",),
    "right": ("center",),
    "outer": ("center",),
    "inner": ("center",),
}


class Axis:
    """A single direction along a model grid, containing potentially multiple cell positions."""

    def __init__(self, ds, name: str, coords: Mapping[str, str], default_shifts: Optional[Mapping[str, str]] = None,
                 padding: Optional[Union[str, Mapping]] = None, fill_value: Optional[float] = None, **kwargs):
        if "boundary" in kwargs:
            raise ValueError("Argument 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")
        if not isinstance(name, str):
            raise TypeError(f"name argument must be of type str, but is of type {type(name)}")
        if not (hasattr(ds, "dims") and hasattr(ds, "coords")):
            raise TypeError(f"ds argument must be of type xarray.Dataset, but is of type {type(ds)}")

        known_dims = set(ds.dims)
        counts: Dict[str, int] = {}
        for pos, dim in coords.items():
            if pos not in POSITIONS:
                raise ValueError(f"Axis position must be one of {list(POSITIONS)}, but got {pos}")
            if dim not in known_dims:
                raise ValueError(
                    f"Could not find dimension `{dim}` (for the `{pos}` position on axis `{name}`) in input dataset."
                )
            counts[dim] = counts.get(dim, 0) + 1
        repeated = sorted(d for d, c in counts.items() if c > 1)
        if repeated:
            raise ValueError(
                f"The same dimension cannot be assigned to multiple positions on axis `{name}`. "
                f"Duplicate dimension(s): {repeated}"
            )

        self._name = name
        self._coords = coords
        self._facedim = None            # set by Grid._assign_face_connections
        self._face_connections = None

        user_shifts = default_shifts or {}
        shifts: Dict[str, str] = {}
        for pos in coords:
            target = user_shifts.get(pos)
            if target is None:
                target = next((cand for cand in FALLBACK_SHIFTS[pos] if cand in coords), None)
            if target is None:
                continue
            if target == pos:
                raise ValueError(f"Can't set the default shift for {pos} to be to {pos}")
            shifts[pos] = target
        self._default_shifts = shifts

        if isinstance(padding, Mapping):
            from .padding import FoldSpec

            padding = FoldSpec.parse(padding)
        elif padding is not None and padding not in VALID_PADDINGS:
            raise ValueError(
                f"padding must be one of {list(VALID_PADDINGS)} "
                f"or a fold spec (e.g. {{'fold': 'corner'}}) or None, but got {padding}"
            )
        self._padding = padding

        if fill_value is None:
            fill_value = 0.0
        if not isinstance(fill_value, (int, float)):
            raise TypeError("fill value must be an integer or a float")
        self._fill_value = fill_value
        self._periodic = padding == "periodic"

    # ---- read-only views ----------------------------------------------------------------
    name = property(lambda self: self._name)
    coords = property(lambda self: self._coords)
    default_shifts = property(lambda self: self._default_shifts)
    padding = property(lambda self: self._padding)
    fill_value = property(lambda self: self._fill_value)
    periodic = property(lambda self: self._periodic)

    @property
    def boundary(self):
        raise AttributeError("Attribute 'boundary' has been renamed to 'padding'. Please use 'padding' instead.")

    def _coord_desc(self):
        lines = []
        for pos, dim in self.coords.items():
            line = "  * %-8s %s" % (pos, dim)
            if pos in self._default_shifts:
                line += " --> %s" % self._default_shifts[pos]
            lines.append(line)
        return lines

    def __repr__(self) -> str:
        head = "<xgcm.Axis '%s' (%s, padding=%r)>" % (
            self.name, "periodic" if self._periodic else "not periodic", self.padding)
        return "\n".join([head, "Axis Coordinates:"] + self._coord_desc())

    # ---- lookups used by the dispatch (index/count work: must be exact) -----------------
    def _get_position_name(self, da) -> Tuple[str, str]:
        """(position, dim) of the single dim of `da` that belongs to this axis."""
        hits = [(pos, dim) for pos, dim in self.coords.items() if dim in da.dims]
        if not hits:
            raise KeyError(f"None of the DataArray's dims {da.dims} were found in axis coords.")
        if len(hits) > 1:
            raise KeyError(f"DataArray cannot have more than 1 axis dimension, but found {set(d for _, d in hits)}")
        return hits[0]

    def _get_axis_dim_num(self, da) -> int:
        _, dim = self._get_position_name(da)
        return da.get_axis_num(dim)
