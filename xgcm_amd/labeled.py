"""Minimal labelled arrays (`DataArray`, `Dataset`) for the xarray-in / xarray-out surface.

xarray is not installable in the target image, so the Grid API is exercised through this
small duck type that implements the subset of `xarray.DataArray` / `xarray.Dataset` the
reference hot path touches (dims, sizes, coords, rename/transpose/isel, name-based
broadcasting arithmetic, sum/cumsum).  `.data` is either a host `numpy.ndarray` or a
HBM-resident `torch.Tensor`; every arithmetic method runs on the GPU through
`xgcm_amd.device` (never numpy) and returns data of the same residency it was given.

When real xarray objects are passed to `Grid` they are converted with `from_xarray` and the
results converted back with `to_xarray` (xgcm_amd/grid.py).
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import device as _dev
from . import dtypes as _dt

try:  # torch is plumbing for device memory only
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore


_LAZY_HOOK = None  # set by xgcm_amd.lazy: (self, other, op, reflexive, dims_order) -> deferred result or None


def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _is_chunked(x) -> bool:
    from .chunked import is_chunked

    return is_chunked(x)


def _shape(x) -> Tuple[int, ...]:
    return tuple(int(s) for s in x.shape)


class DataArray:
    """N-D array with named dimensions and coordinates (subset of xarray.DataArray)."""

    __slots__ = ("data", "dims", "coords", "name", "attrs")
    # numpy interop as xarray's: `np.asarray(da)` / `np.testing.assert_allclose(da, ...)` see the values (HBM data comes
    # to the host for that); `ndarray OP da` is left to the reflected operators below, not broadcast over an object array
    __array_ufunc__ = None

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __init__(self, data, dims: Optional[Sequence[str]] = None, coords=None, name: Optional[str] = None,
                 attrs: Optional[dict] = None):
        if isinstance(data, DataArray):
            dims = data.dims if dims is None else dims
            coords = data.coords if coords is None else coords
            name = data.name if name is None else name
            data = data.data
        if not _is_tensor(data) and not _is_chunked(data):  # (a dask-style chunked host array stays what it is: xgcm_amd.chunked)
            data = np.asarray(data)
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(data.ndim))
        if isinstance(dims, str):
            dims = (dims,)
        dims = tuple(dims)
        if len(dims) != len(data.shape):
            raise ValueError(f"different number of dimensions on data ({len(data.shape)}) and dims {dims}")
        if len(set(dims)) != len(dims):
            raise ValueError(f"duplicate dimension names in {dims}")
        self.data = data
        self.dims = dims
        self.name = name
        self.attrs = dict(attrs) if attrs else {}
        self.coords: "OrderedDict[str, DataArray]" = OrderedDict()
        if coords:
            self._set_coords(coords)

    # ---- construction helpers -----------------------------------------------------------
    def _set_coords(self, coords: Mapping[str, Any]) -> None:
        sizes = self.sizes
        for cname, c in coords.items():
            if isinstance(c, DataArray):
                cda = DataArray(c.data, c.dims, name=cname, attrs=c.attrs)
            elif isinstance(c, tuple) and len(c) >= 2 and isinstance(c[0], (str, list, tuple)):
                cdims = (c[0],) if isinstance(c[0], str) else tuple(c[0])
                cda = DataArray(c[1], cdims, name=cname, attrs=c[2] if len(c) > 2 else None)
            else:
                cda = DataArray(c, (cname,), name=cname)
            for d, s in zip(cda.dims, cda.shape):
                if d not in sizes:
                    raise ValueError(f"coordinate {cname} has dimension {d} not present on the array dims {self.dims}")
                if sizes[d] != s:
                    raise ValueError(
                        f"conflicting sizes for dimension {d!r}: length {s} on {cname!r} and length {sizes[d]} on the data"
                    )
            self.coords[cname] = cda

    def _replace(self, data=None, dims=None, coords=None, name="__keep__") -> "DataArray":
        out = DataArray.__new__(DataArray)
        out.data = self.data if data is None else data
        out.dims = self.dims if dims is None else tuple(dims)
        out.name = self.name if name == "__keep__" else name
        out.attrs = dict(self.attrs)
        out.coords = OrderedDict(self.coords if coords is None else coords)
        return out

    # ---- basic properties ---------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        return _shape(self.data)

    @property
    def ndim(self) -> int:
        return len(self.dims)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def sizes(self) -> Dict[str, int]:
        return dict(zip(self.dims, self.shape))

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def values(self) -> np.ndarray:
        return np.asarray(_dev.tohost(self.data))

    @property
    def is_device(self) -> bool:
        return _is_tensor(self.data) and self.data.is_cuda

    @property
    def chunks(self):
        """block lengths per dim (dask's tuple of tuples) of a chunked host array, else None"""
        if not _is_chunked(self.data):
            return None
        from .chunked import normalize_chunks

        return normalize_chunks(self.data.chunks, self.data.shape)

    def get_axis_num(self, dim: str) -> int:
        try:
            return self.dims.index(dim)
        except ValueError:
            raise ValueError(f"{dim!r} not found in array dimensions {self.dims!r}")

    def to_device(self) -> "DataArray":
        return self._replace(data=_dev.asdevice(self.data))

    def to_host(self) -> "DataArray":
        return self._replace(data=_dev.tohost(self.data))

    # ---- metadata-only transformations --------------------------------------------------
    def copy(self, deep: bool = False, data=None) -> "DataArray":
        if data is not None:
            if _shape(data) != self.shape:
                raise ValueError("replacement data must match the shape")
            return self._replace(data=data if _is_tensor(data) else np.asarray(data))
        if deep:
            d = self.data.clone() if _is_tensor(self.data) else self.data.copy()
            return self._replace(data=d)
        return self._replace()

    def rename(self, new_name_or_name_dict=None, **names) -> "DataArray":
        if new_name_or_name_dict is None or isinstance(new_name_or_name_dict, Mapping):
            mapping = dict(new_name_or_name_dict or {})
            mapping.update(names)
            dims = tuple(mapping.get(d, d) for d in self.dims)
            coords = OrderedDict()
            for cname, c in self.coords.items():
                cd = tuple(mapping.get(d, d) for d in c.dims)
                coords[mapping.get(cname, cname)] = c._replace(dims=cd, name=mapping.get(cname, cname))
            return self._replace(dims=dims, coords=coords)
        return self._replace(name=new_name_or_name_dict)

    def transpose(self, *dims: str) -> "DataArray":
        if not dims:
            dims = self.dims[::-1]
        if set(dims) != set(self.dims) or len(dims) != len(self.dims):
            raise ValueError(f"{dims} must be a permutation of {self.dims}")
        if tuple(dims) == self.dims:
            return self._replace()
        perm = [self.dims.index(d) for d in dims]
        if _is_tensor(self.data):
            data = self.data.permute(*perm)
        else:
            data = np.transpose(self.data, perm)
        return self._replace(data=data, dims=dims)

    @property
    def T(self) -> "DataArray":
        return self.transpose()

    def isel(self, indexers: Optional[Mapping[str, Any]] = None, **kw) -> "DataArray":
        idx = dict(indexers or {})
        idx.update(kw)
        key = []
        dims = []
        for d in self.dims:
            k = idx.get(d, slice(None))
            key.append(k)
            if isinstance(k, slice):
                dims.append(d)
            elif not isinstance(k, (int, np.integer)):
                raise NotImplementedError("isel supports ints and slices only")
        for d in idx:
            if d not in self.dims:
                raise ValueError(f"Dimensions {{{d!r}}} do not exist. Expected one or more of {self.dims}")
        key = tuple(key)
        if _is_tensor(self.data) and any(isinstance(k, slice) and k.step not in (None, 1) for k in key):
            raise NotImplementedError("strided isel on device data")
        data = self.data[key]
        coords = OrderedDict()
        for cname, c in self.coords.items():
            sub = {d: idx[d] for d in c.dims if d in idx}
            if all(isinstance(v, slice) for v in sub.values()):
                coords[cname] = c.isel(sub) if sub else c
            elif len(c.dims) != len(sub):  # dropped dims of a multi-dim coord
                coords[cname] = c.isel(sub)
        return self._replace(data=data, dims=dims, coords=coords)

    def reset_coords(self, names=None, drop: bool = False) -> "DataArray":
        if not drop:
            raise NotImplementedError("reset_coords(drop=False)")
        keep = OrderedDict((k, v) for k, v in self.coords.items() if k in self.dims and v.dims == (k,))
        return self._replace(coords=keep)

    def drop_vars(self, names, errors: str = "raise") -> "DataArray":
        if isinstance(names, str):
            names = [names]
        names = set(names)
        return self._replace(coords=OrderedDict((k, v) for k, v in self.coords.items() if k not in names))

    def assign_coords(self, coords: Optional[Mapping[str, Any]] = None, **kw) -> "DataArray":
        out = self._replace()
        allc = dict(coords or {})
        allc.update(kw)
        out._set_coords(allc)
        return out

    def to_dataset(self, name: Optional[str] = None) -> "Dataset":
        return Dataset({name or self.name: self})

    # ---- comparisons --------------------------------------------------------------------
    def equals(self, other: "DataArray") -> bool:
        if is_xarray(other) and type(other).__name__ == "DataArray":
            other = from_xarray(other)
        if not isinstance(other, DataArray) or self.dims != other.dims or self.shape != other.shape:
            return False
        if not np.array_equal(self.values, other.values, equal_nan=True):
            return False
        if set(self.coords) != set(other.coords):
            return False
        return all(self.coords[k].dims == other.coords[k].dims
                   and np.array_equal(self.coords[k].values, other.coords[k].values) for k in self.coords)

    def identical(self, other: "DataArray") -> bool:
        return self.equals(other) and self.name == other.name

    # ---- arithmetic on the GPU (name-based broadcasting like xarray) ---------------------
    def _binary(self, other, op: str, reflexive: bool = False, dims_order: Optional[Sequence[str]] = None) -> "DataArray":
        """`self OP other` with name-based broadcasting.  `dims_order` (internal): lay the result out with its dims in
        that order instead of xarray's (self's dims, then other's new ones) -- same values, no transposed copy later."""
        if is_xarray(other):  # a deferred result (or any array of ours) next to a real xarray object: ours, by name
            other = from_xarray(other)
        if _LAZY_HOOK is not None:  # `array OP deferred_result`, `field * metric` inside `grid.fused()`: xgcm_amd.lazy
            res = _LAZY_HOOK(self, other, op, reflexive, dims_order)
            if res is not None:
                return res
        if isinstance(other, (int, float, np.integer, np.floating)):
            # python scalars are "weak" (numpy's promotion): a float32 array stays float32 next to 2.0, an integer array
            # stays integral next to 2 and becomes float64 next to 2.0; numpy scalars carry their own dtype
            sdt = np.result_type(_dt.np_dtype(self.data), other)
            o = DataArray(np.full((1,) * self.ndim, other, dtype=sdt), tuple(f"__s{i}" for i in range(self.ndim)))
            a, b, dims = self.data, o.data, self.dims
            coords = OrderedDict(self.coords)
        elif isinstance(other, DataArray):
            dims = self.dims + tuple(d for d in other.dims if d not in self.dims)
            if dims_order is not None:
                dims = tuple(d for d in dims_order if d in dims) + tuple(d for d in dims if d not in dims_order)
            a = _aligned_view(self, dims)
            b = _aligned_view(other, dims)
            for d in dims:
                if d in self.dims and d in other.dims and self.sizes[d] != other.sizes[d]:
                    raise ValueError(f"cannot broadcast: dimension {d!r} has sizes {self.sizes[d]} and {other.sizes[d]}")
            coords = OrderedDict(self.coords)
            for k, v in other.coords.items():
                mine = coords.get(k)
                if mine is None:
                    coords[k] = v
                elif k not in dims and not _same_coord(mine, v):
                    # xarray: "other [than index] coordinates are not aligned, and if their values conflict, they will be
                    # dropped" (`arr[0] - arr[1]` loses its scalar coordinate)
                    del coords[k]
        elif (isinstance(other, np.ndarray) or _is_tensor(other)) and other.ndim <= self.ndim:
            # an unlabelled array: numpy's positional broadcasting against this array's shape, as in xarray
            shape = (1,) * (self.ndim - other.ndim) + tuple(int(n) for n in other.shape)
            if any(m not in (1, n) for m, n in zip(shape, self.shape)):
                raise ValueError(f"operands could not be broadcast together with shapes {self.shape} {tuple(other.shape)}")
            a, b, dims = self.data, other.reshape(shape), self.dims
            coords = OrderedDict(self.coords)
        else:
            return NotImplemented
        if reflexive:
            a, b = b, a
        host = not (_is_tensor(self.data) or _is_tensor(other) or (isinstance(other, DataArray) and _is_tensor(other.data)))
        res = _dev.binary(op, a, b)
        if host:
            res = _dev.tohost(res)
        return DataArray(res, dims, coords=coords, name=_result_name(self, other))

    def __mul__(self, o): return self._binary(o, "mul")
    def __rmul__(self, o): return self._binary(o, "mul", True)
    def __truediv__(self, o): return self._binary(o, "div")
    def __rtruediv__(self, o): return self._binary(o, "div", True)
    def __add__(self, o): return self._binary(o, "add")
    def __radd__(self, o): return self._binary(o, "add", True)
    def __sub__(self, o): return self._binary(o, "sub")
    def __rsub__(self, o): return self._binary(o, "sub", True)

    def __neg__(self):
        return self._binary(-1.0 if np.dtype(self.dtype).kind == "f" else -1, "mul", True)

    def __abs__(self):
        data = self.data
        return self._replace(data=data.abs() if _is_tensor(data) else np.abs(data))

    def __getattr__(self, key: str):
        """`da.time`: coordinates as attributes, as xarray allows"""
        if not key.startswith("_"):
            try:
                coords = object.__getattribute__(self, "coords")
            except AttributeError:
                coords = None
            if coords and key in coords:
                return coords[key]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {key!r}")

    def sum(self, dim=None, skipna: Optional[bool] = None, keep_attrs: bool = False, **kwargs) -> "DataArray":
        """Sum over `dim` (str or list); float default skips NaN like xarray."""
        if kwargs:
            raise TypeError(f"sum() got unexpected keyword argument(s): {list(kwargs)}")
        dims = list(self.dims) if dim is None else ([dim] if isinstance(dim, str) else list(dim))
        out = self
        host = not _is_tensor(self.data)
        data = _dev.asdevice(self.data)
        cur_dims = list(self.dims)
        for d in dims:
            ax = cur_dims.index(d)
            data = _dev.reduce1d(data, ax, None, True if skipna is None else bool(skipna))
            cur_dims.pop(ax)
        coords = OrderedDict((k, v) for k, v in out.coords.items() if all(cd in cur_dims for cd in v.dims))
        return DataArray(_dev.tohost(data) if host else data, cur_dims, coords=coords, name=self.name,
                         attrs=self.attrs if keep_attrs else None)

    def cumsum(self, dim: str, skipna: Optional[bool] = None) -> "DataArray":
        host = not _is_tensor(self.data)
        ax = self.get_axis_num(dim)
        res = _dev.cumsum1d(self.data, ax, 0, 0, 0, 0, None, 0.0, False, True if skipna is None else bool(skipna))
        return self._replace(data=_dev.tohost(res) if host else res)

    def __repr__(self) -> str:
        where = "HBM" if self.is_device else "host"
        return f"<xgcm_amd.DataArray {self.name!r} {dict(self.sizes)} [{where}] coords={list(self.coords)}>"

    def __len__(self) -> int:
        return self.shape[0]

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        if not isinstance(key, tuple):
            key = (key,)
        return self.isel({d: k for d, k in zip(self.dims, key)})


def _same_coord(a, b) -> bool:
    return a.dims == b.dims and a.shape == b.shape and bool(np.array_equal(a.values, b.values))


def _result_name(a, b):
    """xarray's rule for `a OP b`: the name survives only if every named operand carries the same one (a scalar has none
    to disagree with) -- so `grid.diff(T, "X") / dxC` is nameless, as in the reference (xgcm/grid.py:1576-1578)"""
    if isinstance(b, DataArray):
        return a.name if a.name == b.name else None
    return a.name


def _aligned_view(da: DataArray, dims: Sequence[str]):
    """View of `da.data` with exactly `dims` (size-1 where da lacks the dim); no copy on device."""
    present = [d for d in dims if d in da.dims]
    perm = [da.dims.index(d) for d in present]
    data = da.data
    if _is_tensor(data):
        v = data.permute(*perm) if perm != list(range(len(perm))) else data
        index = tuple(slice(None) if d in da.dims else None for d in dims)
        return v[index]
    if _is_chunked(data):
        # a chunked host array stays chunked under name-based broadcasting: a view with its dims reordered / extended by name
        if len(present) == len(dims) and perm == list(range(len(perm))):
            return data
        from .chunked import ExpandedView

        return ExpandedView(data, [da.dims.index(d) if d in da.dims else None for d in dims])
    v = np.transpose(data, perm) if perm != list(range(len(perm))) else data
    index = tuple(slice(None) if d in da.dims else np.newaxis for d in dims)
    return v[index]


class Dataset:
    """Dict of named DataArrays sharing dimensions (subset of xarray.Dataset)."""

    def __init__(self, data_vars: Optional[Mapping[str, Any]] = None, coords: Optional[Mapping[str, Any]] = None,
                 attrs: Optional[dict] = None):
        self.data_vars: "OrderedDict[str, DataArray]" = OrderedDict()
        self.coords: "OrderedDict[str, DataArray]" = OrderedDict()
        self.attrs = dict(attrs) if attrs else {}
        self._sizes: Dict[str, int] = {}
        # data variables of a converted xarray.Dataset stay where they are until somebody asks for one (`from_xarray`):
        # a Grid needs coordinates and a handful of metrics, not the model output
        self._lazy: "OrderedDict[str, Tuple[dict, Any]]" = OrderedDict()  # name -> (attrs, loader)
        for k, v in (coords or {}).items():
            self._add(k, v, True)
        for k, v in (data_vars or {}).items():
            self._add(k, v, False)

    def _as_da(self, name, v) -> DataArray:
        if isinstance(v, DataArray):
            return v._replace(name=name)
        if isinstance(v, tuple):
            dims = (v[0],) if isinstance(v[0], str) else tuple(v[0])
            return DataArray(v[1], dims, name=name, attrs=v[2] if len(v) > 2 else None)
        return DataArray(v, (name,), name=name)

    def _add(self, name, v, is_coord: bool) -> None:
        da = self._as_da(name, v)
        for d, s in da.sizes.items():
            if self._sizes.setdefault(d, s) != s:
                raise ValueError(f"conflicting sizes for dimension {d!r}: length {s} on {name!r} and length {self._sizes[d]}")
        if is_coord:
            self.coords[name] = da._replace(coords=OrderedDict())
        else:
            for cn, c in da.coords.items():
                if cn not in self.coords:
                    self._add(cn, c, True)
            self.data_vars[name] = da._replace(coords=OrderedDict())

    @property
    def dims(self) -> Dict[str, int]:
        return dict(self._sizes)

    sizes = dims

    def _load(self, key: str) -> None:
        attrs, loader = self._lazy.pop(key)
        self._add(key, loader(), False)

    @property
    def variables(self) -> Dict[str, DataArray]:
        for key in list(self._lazy):
            self._load(key)
        out = OrderedDict(self.coords)
        out.update(self.data_vars)
        return out

    def variable_attrs(self) -> Dict[str, dict]:
        """attrs of every variable, without loading the data of those still waiting in the source dataset"""
        out = OrderedDict((k, c.attrs) for k, c in self.coords.items())
        out.update((k, v.attrs) for k, v in self.data_vars.items())
        out.update((k, a) for k, (a, _) in self._lazy.items())
        return out

    def __contains__(self, key) -> bool:
        return key in self.data_vars or key in self.coords or key in self._lazy

    def __getitem__(self, key: str) -> DataArray:
        if key in self._lazy:
            self._load(key)
        if key in self.data_vars:
            base = self.data_vars[key]
        elif key in self.coords:
            base = self.coords[key]
        else:
            raise KeyError(key)
        coords = OrderedDict((k, c) for k, c in self.coords.items() if all(d in base.dims for d in c.dims))
        return base._replace(coords=coords, name=key)

    def __setitem__(self, key: str, value) -> None:
        self._add(key, value, False)

    def __getattr__(self, key: str):
        if key.startswith("_") or key in ("data_vars", "coords", "attrs"):
            raise AttributeError(key)
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def copy(self) -> "Dataset":
        out = Dataset(attrs=self.attrs)
        out.coords = OrderedDict(self.coords)
        out.data_vars = OrderedDict(self.data_vars)
        out._sizes = dict(self._sizes)
        out._lazy = OrderedDict(self._lazy)
        return out

    def __repr__(self) -> str:
        return f"<xgcm_amd.Dataset dims={self._sizes} data_vars={list(self.data_vars) + list(self._lazy)} coords={list(self.coords)}>"


# ---- optional bridges to real xarray ------------------------------------------------------
def is_xarray(obj) -> bool:
    """an object of the package named `xarray` -- or a deferred result that stands for one (`lazy.LazyArray` born of xarray
    inputs: it comes back in as xarray came in, and what is computed from it goes out as xarray)"""
    mod = type(obj).__module__ or ""
    return mod.startswith("xarray") or (type(obj).__name__ == "LazyArray" and bool(getattr(obj, "_xr", False)))


CHUNKED_INPUT_MESSAGE = (  # what the chunked path does not serve (connected topologies, vector components, user ufuncs)
    "this call does not take dask-chunked inputs on the MI355X backend: compute the array first, or hand its blocks "
    "to xgcm_amd.streaming.stream_records (record blocks streamed through HBM)"
)


def from_xarray(obj):
    """xarray.DataArray / Dataset -> xgcm_amd labelled object (host data).  A dask-backed DataArray keeps its dask array as
    `.data` -- nothing is computed here; the operators walk its blocks (xgcm_amd.chunked; reference: `dask="parallelized"`,
    grid.py:786-818)."""
    tname = type(obj).__name__
    if tname == "LazyArray":  # a deferred result standing for an xarray object: the same deferred value, as one of ours
        out = obj._replace()
        out._xr = False
        return out
    if tname == "DataArray" and getattr(obj, "chunks", None) is not None:
        coords = {k: (tuple(v.dims), np.asarray(v.values), dict(v.attrs)) for k, v in obj.coords.items()}
        return DataArray(obj.data, tuple(obj.dims), coords=coords, name=obj.name, attrs=dict(obj.attrs))
    if tname == "DataArray":
        coords = {k: (tuple(v.dims), np.asarray(v.values), dict(v.attrs)) for k, v in obj.coords.items()}
        return DataArray(np.asarray(obj.values), tuple(obj.dims), coords=coords, name=obj.name, attrs=dict(obj.attrs))
    if tname == "Dataset":
        # coordinates now; data variables only when asked for (`Grid(ds)` of a model run must not read the run): their
        # names, dims and attrs are known at once, their values are fetched by `ds[name]`
        coords = {k: (tuple(v.dims), np.asarray(v.values), dict(v.attrs)) for k, v in obj.coords.items()}
        out = Dataset(None, coords, attrs=dict(obj.attrs))
        for k, v in obj.data_vars.items():
            dims, shape = tuple(v.dims), tuple(getattr(v, "shape", ()))
            for d, n in zip(dims, shape):
                if out._sizes.setdefault(d, int(n)) != int(n):
                    raise ValueError(f"conflicting sizes for dimension {d!r}: length {n} on {k!r} and length {out._sizes[d]}")
            out._lazy[k] = (dict(v.attrs), (lambda v=v, k=k: DataArray(np.asarray(v.values), tuple(v.dims), name=k, attrs=dict(v.attrs))))
        return out
    raise TypeError(type(obj))


def to_xarray(da: DataArray):
    import xarray as xr  # only reached when the caller handed us xarray objects

    # coordinate variables leave as they came in: dims, values in their own dtype AND attrs (the reference takes them
    # from `grid._ds` unchanged, xgcm/grid_ufunc.py:1262-1320)
    coords = {k: (c.dims, c.values, dict(c.attrs)) for k, c in da.coords.items()}
    data = None
    if _is_chunked(da.data):  # a chunked result leaves as a dask array of the same blocks where dask exists (else: assembled)
        try:
            data = da.data.to_dask()
        except Exception:  # noqa: BLE001 -- no dask for this interpreter, or a container that is not a BlockArray
            data = None
    if data is None:
        data = da.values
    return xr.DataArray(data, dims=da.dims, coords=coords, name=da.name, attrs=da.attrs)
