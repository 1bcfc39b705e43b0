"""A numpy-backed stand-in for the part of xarray the reference's Grid stack touches.

TEST INFRASTRUCTURE -- used by `oracle/make_golden_grid.py` in the build container only (never shipped, never imported by
the product, never on the GPU box).  xarray is not installable in this image, so `import xgcm` fails; with this module
registered under the name `xarray` (and an empty `dask.array.Array`) the reference's OWN `xgcm/grid.py`, `axis.py`,
`grid_ufunc.py`, `padding.py`, `gridops.py`, `metrics.py` import and run unmodified, and their results on seeded datasets
become golden fixtures for `xgcm_amd.Grid` (tests/golden/grid_reference.*).

What such fixtures pin: the reference's LOGIC -- signature dispatch, per-axis kwargs, the `Grid.cumsum` trim / pad table,
metric selection and interpolation, `derivative` / `integrate` / `average` / `cumint`, coordinate re-attachment and dim
order -- under THIS module's semantics of the container operations.  "Pinned modulo stand-ins": a difference between this
file and real xarray is not caught.  The semantics implemented here are the documented xarray ones for exactly the calls
the reference makes:

* DataArray: dims / shape / sizes / dtype / name / attrs / data / values / coords / chunks (None); name-based broadcasting
  arithmetic (dims of the left operand first, then new ones; coordinates of both operands kept, left wins);
  transpose (with ...), rename, isel (slices / ints), pad (numpy.pad on the data, coordinates along padded dims dropped),
  cumsum / sum (float default skipna=True -> nancumsum / nansum), weighted(w).mean (sum(x*w) / sum(w where x valid)),
  reset_coords(drop=True), reset_index(drop=True), drop_vars, assign_coords, copy, squeeze, expand_dims, astype;
* Dataset: variables by name with the coordinates that fit their dims; dims / sizes / coords / data_vars / variables;
* apply_ufunc: core dims moved last, `exclude_dims` may change length, outputs carry the broadcast dims followed by the
  output core dims and the input coordinates that do not touch an excluded dim;
* concat along an existing or new dim.
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Hashable, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np


def _as_dims(dims) -> Tuple[str, ...]:
    if dims is None:
        return ()
    if isinstance(dims, str):
        return (dims,)
    return tuple(dims)


class _Coords(Mapping):
    """`da.coords` / `ds.coords`: name -> DataArray (the coordinate variable with the coordinates that fit ITS dims)"""

    def __init__(self, owner):
        self._o = owner

    def __getitem__(self, key):
        dims, values, attrs = self._o._coords[key]
        sub = OrderedDict((k, v) for k, v in self._o._coords.items() if set(v[0]) <= set(dims))
        return DataArray(values, coords=None, dims=dims, name=key, attrs=attrs, _raw_coords=sub)

    def __iter__(self):
        return iter(self._o._coords)

    def __len__(self):
        return len(self._o._coords)

    def __contains__(self, key):
        return key in self._o._coords

    def to_dataset(self):
        return Dataset(coords={k: self[k] for k in self})


def _coord_tuple(name, value, owner_sizes=None) -> Tuple[Tuple[str, ...], np.ndarray, dict]:
    """(dims, values, attrs) of one coordinate given as a DataArray, a (dims, values[, attrs]) tuple or a bare array"""
    if isinstance(value, DataArray):
        return value.dims, np.asarray(value.data), dict(value.attrs)
    if isinstance(value, tuple) and len(value) >= 2 and isinstance(value[0], (str, list, tuple)):
        return _as_dims(value[0]), np.asarray(value[1]), dict(value[2]) if len(value) > 2 and value[2] else {}
    arr = np.asarray(value)
    return ((name,) if arr.ndim else ()), arr, {}


class DataArray:
    __array_priority__ = 50

    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None, _raw_coords=None):
        if isinstance(data, DataArray):
            coords = coords if coords is not None else data._coords
            dims = dims if dims is not None else data.dims
            name = name if name is not None else data.name
            attrs = attrs if attrs is not None else data.attrs
            data = data.data
        self.data = np.asarray(data)
        if dims is None:
            if isinstance(coords, (list, tuple)) and coords and all(isinstance(c, tuple) for c in coords):
                dims = [c[0] for c in coords]
                coords = {c[0]: c[1] for c in coords}
            elif isinstance(coords, Mapping) and len(coords) == self.data.ndim and self.data.ndim:
                dims = list(coords)
            else:
                dims = [f"dim_{i}" for i in range(self.data.ndim)]
        self.dims = _as_dims(dims)
        if len(self.dims) != self.data.ndim:
            raise ValueError(f"different number of dimensions on data and dims: {self.data.ndim} vs {len(self.dims)}")
        self.name = name
        self.attrs = dict(attrs) if attrs else {}
        self._coords: "OrderedDict[str, Tuple]" = OrderedDict()
        if _raw_coords is not None:
            self._coords.update(_raw_coords)
        elif coords is not None:
            items = coords.items() if isinstance(coords, Mapping) else coords
            for k, v in items:
                self._set_coord(k, v)

    # ---- basics ----------------------------------------------------------------------------
    def _set_coord(self, name, value):
        dims, values, attrs = _coord_tuple(name, value)
        sizes = self.sizes
        for d, n in zip(dims, values.shape):
            if d not in sizes:
                raise ValueError(f"coordinate {name} has dimensions {dims}, but these are not a subset of the DataArray dimensions {self.dims}")
            if sizes[d] != n:
                raise ValueError(f"conflicting sizes for dimension {d!r}: length {n} on {name!r} and length {sizes[d]} on the data")
        self._coords[name] = (dims, values, attrs)

    @property
    def shape(self):
        return tuple(int(s) for s in self.data.shape)

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def size(self):
        return int(self.data.size)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def sizes(self):
        return OrderedDict(zip(self.dims, self.shape))

    @property
    def values(self):
        return np.asarray(self.data)

    @property
    def coords(self):
        return _Coords(self)

    @property
    def chunks(self):
        return None

    @property
    def variable(self):
        return self

    def get_axis_num(self, dim):
        if dim not in self.dims:
            raise ValueError(f"{dim!r} not found in array dimensions {self.dims!r}")
        return self.dims.index(dim)

    def __getitem__(self, key):
        if isinstance(key, str):
            if key not in self._coords and key in self.dims:  # a dim without a coordinate: xarray hands out its range index
                return DataArray(np.arange(self.sizes[key]), dims=(key,), name=key)
            return self.coords[key]
        raise NotImplementedError("positional indexing: use isel")

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"<xr_min.DataArray {self.name!r} {dict(self.sizes)} coords={list(self._coords)}>"

    def _new(self, data, dims, coords=None, name="__keep__", attrs="__keep__"):
        out = DataArray.__new__(DataArray)
        out.data = np.asarray(data)
        out.dims = _as_dims(dims)
        out.name = self.name if name == "__keep__" else name
        out.attrs = dict(self.attrs) if attrs == "__keep__" else dict(attrs or {})
        out._coords = OrderedDict(self._coords if coords is None else coords)
        return out

    def copy(self, deep=True, data=None):
        d = self.data if data is None else np.asarray(data)
        return self._new(d.copy() if (deep and data is None) else d, self.dims)

    def astype(self, dtype, **kw):
        return self._new(self.data.astype(dtype), self.dims)

    # ---- coordinates -----------------------------------------------------------------------
    def reset_coords(self, names=None, drop=False):
        if not drop:
            raise NotImplementedError("reset_coords(drop=False)")
        keep = OrderedDict((k, v) for k, v in self._coords.items() if v[0] == (k,) and k in self.dims)  # index coords stay
        return self._new(self.data, self.dims, keep)

    def reset_index(self, dims_or_levels, drop=False):
        if not drop:
            raise NotImplementedError("reset_index(drop=False)")
        names = [dims_or_levels] if isinstance(dims_or_levels, str) else list(dims_or_levels)
        return self._new(self.data, self.dims, OrderedDict((k, v) for k, v in self._coords.items() if k not in names))

    def drop_vars(self, names, errors="raise"):
        names = [names] if isinstance(names, str) else list(names)
        return self._new(self.data, self.dims, OrderedDict((k, v) for k, v in self._coords.items() if k not in names))

    def assign_coords(self, coords=None, **kw):
        out = self._new(self.data, self.dims)
        for k, v in dict(coords or {}, **kw).items():
            out._set_coord(k, v)
        return out

    # ---- shape -----------------------------------------------------------------------------
    def transpose(self, *dims, **kw):
        if not dims:
            dims = self.dims[::-1]
        if Ellipsis in dims:
            named = [d for d in dims if d is not Ellipsis]
            rest = [d for d in self.dims if d not in named]
            i = dims.index(Ellipsis)
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        if set(dims) != set(self.dims):
            raise ValueError(f"{dims} must be a permuted list of {self.dims}, unless `...` is included")
        perm = [self.dims.index(d) for d in dims]
        return self._new(np.transpose(self.data, perm), dims)

    def rename(self, new_name_or_name_dict=None, **names):
        if new_name_or_name_dict is None or isinstance(new_name_or_name_dict, Mapping):
            mp = dict(new_name_or_name_dict or {}, **names)
            dims = tuple(mp.get(d, d) for d in self.dims)
            coords = OrderedDict()
            for k, (cd, cv, ca) in self._coords.items():
                coords[mp.get(k, k)] = (tuple(mp.get(d, d) for d in cd), cv, ca)
            return self._new(self.data, dims, coords)
        return self._new(self.data, self.dims, name=new_name_or_name_dict)

    def isel(self, indexers=None, drop=False, **kw):
        idx = dict(indexers or {}, **kw)
        for d in idx:
            if d not in self.dims:
                raise ValueError(f"Dimensions {{{d!r}}} do not exist. Expected one or more of {self.dims}")
        key = tuple(idx.get(d, slice(None)) for d in self.dims)
        dims = tuple(d for d in self.dims if not isinstance(idx.get(d, slice(None)), (int, np.integer)))
        coords = OrderedDict()
        for k, (cd, cv, ca) in self._coords.items():
            ck = tuple(idx.get(d, slice(None)) for d in cd)
            nd = tuple(d for d in cd if not isinstance(idx.get(d, slice(None)), (int, np.integer)))
            coords[k] = (nd, cv[ck], ca)
        return self._new(self.data[key], dims, coords)

    def squeeze(self, dim=None, drop=False):
        dims = [dim] if isinstance(dim, str) else (list(dim) if dim is not None else [d for d, n in self.sizes.items() if n == 1])
        return self.isel({d: 0 for d in dims})

    def expand_dims(self, dim, axis=0):
        dims = [dim] if isinstance(dim, str) else list(dim)
        out = self
        for d in reversed(dims):
            out = out._new(np.expand_dims(out.data, axis), out.dims[:axis] + (d,) + out.dims[axis:])
        return out

    def pad(self, pad_width=None, mode="constant", constant_values=None, **kw):
        pw = dict(pad_width or {}, **{k: v for k, v in kw.items() if isinstance(v, (tuple, list))})
        widths = [tuple(pw.get(d, (0, 0))) for d in self.dims]
        extra = {}
        if mode == "constant":
            # xarray: float arrays default to NaN, an explicit value is cast by numpy.pad to the array's dtype
            extra["constant_values"] = np.nan if constant_values is None else constant_values
        data = np.pad(self.data, widths, mode=mode, **extra)
        coords = OrderedDict((k, v) for k, v in self._coords.items() if not (set(v[0]) & set(pw)))
        return self._new(data, self.dims, coords)

    # ---- arithmetic ------------------------------------------------------------------------
    def _binary(self, other, f, reflexive=False):
        if isinstance(other, DataArray):
            dims = self.dims + tuple(d for d in other.dims if d not in self.dims)
            for d in dims:
                if d in self.dims and d in other.dims and self.sizes[d] != other.sizes[d]:
                    raise ValueError(f"cannot align: dimension {d!r} has sizes {self.sizes[d]} and {other.sizes[d]}")
            a = _aligned(self, dims)
            b = _aligned(other, dims)
            coords = OrderedDict(self._coords)
            for k, v in other._coords.items():
                mine = coords.get(k)
                if mine is None:
                    coords[k] = v
                elif k not in dims and not (mine[0] == v[0] and np.shape(mine[1]) == np.shape(v[1]) and np.array_equal(mine[1], v[1])):
                    del coords[k]  # xarray: non-index coordinates whose values conflict are dropped by arithmetic
            name = self.name if self.name == other.name else None
        elif isinstance(other, Dataset):
            return NotImplemented
        else:
            dims, a, b, coords, name = self.dims, self.data, other, OrderedDict(self._coords), self.name
        res = f(b, a) if reflexive else f(a, b)
        return self._new(res, dims, coords, name=name, attrs={})

    def __mul__(self, o): return self._binary(o, np.multiply)
    def __rmul__(self, o): return self._binary(o, np.multiply, True)
    def __truediv__(self, o): return self._binary(o, np.true_divide)
    def __rtruediv__(self, o): return self._binary(o, np.true_divide, True)
    def __add__(self, o): return self._binary(o, np.add)
    def __radd__(self, o): return self._binary(o, np.add, True)
    def __sub__(self, o): return self._binary(o, np.subtract)
    def __rsub__(self, o): return self._binary(o, np.subtract, True)
    def __neg__(self): return self._new(-self.data, self.dims)

    # ---- reductions ------------------------------------------------------------------------
    def _skip(self, skipna):
        return (self.data.dtype.kind == "f") if skipna is None else bool(skipna)

    def cumsum(self, dim=None, skipna=None, **kw):
        ax = self.get_axis_num(dim if isinstance(dim, str) else list(dim)[0])
        f = np.nancumsum if self._skip(skipna) else np.cumsum
        return self._new(f(self.data, axis=ax), self.dims, attrs={})

    def sum(self, dim=None, skipna=None, keep_attrs=False, **kw):
        dims = list(self.dims) if dim is None else ([dim] if isinstance(dim, str) else list(dim))
        axes = tuple(self.get_axis_num(d) for d in dims)
        f = np.nansum if self._skip(skipna) else np.sum
        out_dims = tuple(d for d in self.dims if d not in dims)
        coords = OrderedDict((k, v) for k, v in self._coords.items() if not (set(v[0]) & set(dims)))
        return self._new(f(self.data, axis=axes), out_dims, coords, attrs=self.attrs if keep_attrs else {})

    def weighted(self, weights):
        return _Weighted(self, weights)

    def notnull(self):
        return self._new(~np.isnan(self.data) if self.data.dtype.kind == "f" else np.ones(self.shape, bool), self.dims)

    def fillna(self, value):
        return self._new(np.where(np.isnan(self.data), value, self.data), self.dims)

    def equals(self, other):
        return isinstance(other, DataArray) and self.dims == other.dims and np.array_equal(self.data, other.data, equal_nan=True)


def _aligned(da: DataArray, dims) -> np.ndarray:
    present = [d for d in dims if d in da.dims]
    v = np.transpose(da.data, [da.dims.index(d) for d in present])
    return v[tuple(slice(None) if d in da.dims else np.newaxis for d in dims)]


class _Weighted:
    """`da.weighted(w)`: xarray's weighted mean = sum(da * w, skipna) / sum(w where da is valid)"""

    def __init__(self, da, weights):
        self.da, self.w = da, weights

    def mean(self, dim=None, skipna=None, keep_attrs=False, **kw):
        da, w = self.da, self.w
        skip = da._skip(skipna)
        num = (da * w).sum(dim, skipna=skip)
        mask = da.notnull() if skip else da._new(np.ones(da.shape, bool), da.dims)
        den = (mask * w).sum(dim, skipna=False)
        out = num / den
        out.data = np.where(den.data == 0, np.nan, out.data) if np.ndim(out.data) else out.data
        out.name = da.name  # (xarray runs the weighted reduction through a temporary dataset: the array keeps ITS name)
        return out

    def sum(self, dim=None, skipna=None, **kw):
        out = (self.da * self.w).sum(dim, skipna=self.da._skip(skipna))
        out.name = self.da.name
        return out


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._vars: "OrderedDict[str, Tuple]" = OrderedDict()    # name -> (dims, values, attrs)
        self._coords: "OrderedDict[str, Tuple]" = OrderedDict()
        self.attrs = dict(attrs) if attrs else {}
        for k, v in (coords.items() if coords else []):
            self._coords[k] = _coord_tuple(k, v)
        for k, v in (data_vars.items() if data_vars else []):
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, DataArray):
            for ck, cv in value._coords.items():
                self._coords.setdefault(ck, cv)
        dims, values, attrs = _coord_tuple(key, value)
        sizes = self.sizes
        for d, n in zip(dims, values.shape):
            if d in sizes and sizes[d] != n:
                raise ValueError(f"conflicting sizes for dimension {d!r}: length {n} on {key!r} and length {sizes[d]}")
        if key in self._coords:
            self._coords[key] = (dims, values, attrs)
        else:
            self._vars[key] = (dims, values, attrs)

    @property
    def sizes(self):
        out = OrderedDict()
        for dims, values, _ in list(self._coords.values()) + list(self._vars.values()):
            for d, n in zip(dims, values.shape):
                out.setdefault(d, int(n))
        return out

    dims = sizes

    @property
    def coords(self):
        return _Coords(self)

    @property
    def data_vars(self):
        return OrderedDict((k, self[k]) for k in self._vars)

    @property
    def variables(self):
        out = OrderedDict((k, self.coords[k]) for k in self._coords)
        out.update(self.data_vars)
        return out

    def __contains__(self, key):
        return key in self._vars or key in self._coords

    def __iter__(self):
        return iter(self._vars)

    def keys(self):
        return self._vars.keys()

    def __getitem__(self, key):
        if key in self._vars:
            dims, values, attrs = self._vars[key]
        elif key in self._coords:
            dims, values, attrs = self._coords[key]
        else:
            raise KeyError(key)
        sub = OrderedDict((k, v) for k, v in self._coords.items() if set(v[0]) <= set(dims))
        return DataArray(values, dims=dims, name=key, attrs=attrs, _raw_coords=sub)

    def __getattr__(self, key):
        if key.startswith("_"):
            raise AttributeError(key)
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def copy(self, deep=False):
        out = Dataset(attrs=self.attrs)
        out._vars, out._coords = OrderedDict(self._vars), OrderedDict(self._coords)
        return out


def concat(objs: Sequence[DataArray], dim: str, **kw) -> DataArray:
    objs = list(objs)
    first = objs[0]
    if dim in first.dims:
        ax = first.dims.index(dim)
        data = np.concatenate([np.transpose(o.data, [o.dims.index(d) for d in first.dims]) for o in objs], axis=ax)
        dims = first.dims
    else:
        data = np.stack([np.transpose(o.data, [o.dims.index(d) for d in first.dims]) for o in objs], axis=0)
        dims = (dim,) + first.dims
    coords = OrderedDict((k, v) for k, v in first._coords.items() if dim not in v[0])
    return first._new(data, dims, coords)


def apply_ufunc(func, *args, input_core_dims=None, output_core_dims=((),), exclude_dims=frozenset(), dask="forbidden",
                output_dtypes=None, dask_gufunc_kwargs=None, kwargs=None, keep_attrs=False, vectorize=False, join="exact",
                **ignored):
    """xarray.apply_ufunc for numpy-backed DataArrays: each argument's core dims are moved LAST (in the order given), the
    remaining dims are broadcast by name (first appearance order), `func` is called on the raw arrays, and every output is
    (broadcast dims ..., its output core dims).  Coordinates of the inputs that do not touch an excluded dim are kept."""
    if input_core_dims is None:
        input_core_dims = [()] * len(args)
    excl = set(exclude_dims)
    bdims = []
    for a, core in zip(args, input_core_dims):
        if isinstance(a, DataArray):
            for d in a.dims:
                if d not in core and d not in bdims:
                    bdims.append(d)
    bsizes = {}
    for a, core in zip(args, input_core_dims):
        if isinstance(a, DataArray):
            for d, n in a.sizes.items():
                if d in core:
                    continue
                if bsizes.setdefault(d, n) != n:
                    raise ValueError(f"operands could not be broadcast along {d!r}")
    raw = []
    for a, core in zip(args, input_core_dims):
        if not isinstance(a, DataArray):
            raw.append(a)
            continue
        missing = [d for d in core if d not in a.dims]
        if missing:
            raise ValueError(f"operand to apply_ufunc has required core dimensions {list(core)}, but some of these dimensions are absent: {missing}")
        order = [d for d in bdims if d in a.dims] + list(core)
        v = np.transpose(a.data, [a.dims.index(d) for d in order])
        # like xarray (computation.py `broadcast_compat_data`): a size-1 axis for a broadcast dim the argument lacks -- except
        # LEADING ones, which numpy's broadcasting supplies itself (a 1-D `target` stays 1-D for the function)
        index = []
        for d in bdims:
            if d in a.dims:
                index.append(slice(None))
            elif index:
                index.append(np.newaxis)
        raw.append(v[tuple(index) + (slice(None),) * len(core)])
    res = func(*raw, **(kwargs or {}))
    outs = res if isinstance(res, tuple) else (res,)
    if len(outs) != len(output_core_dims):
        raise ValueError(f"applied function returned {len(outs)} outputs, expected {len(output_core_dims)}")
    coords = OrderedDict()
    for a in args:
        if isinstance(a, DataArray):
            for k, v in a._coords.items():
                if not (set(v[0]) & excl):
                    coords.setdefault(k, v)
    first = next((a for a in args if isinstance(a, DataArray)), None)
    results = []
    for o, core in zip(outs, output_core_dims):
        o = np.asarray(o)
        dims = tuple(bdims) + tuple(core)
        if o.ndim != len(dims):
            raise ValueError(f"applied function returned data with {o.ndim} dims, expected {len(dims)}: {dims}")
        keep = OrderedDict((k, v) for k, v in coords.items() if set(v[0]) <= set(dims))
        results.append(first._new(o, dims, keep, name=first.name if all(getattr(a, "name", first.name) == first.name for a in args if isinstance(a, DataArray)) else None,
                                  attrs=first.attrs if keep_attrs else {}))
    return tuple(results) if len(results) > 1 else results[0]
