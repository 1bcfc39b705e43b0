"""More of the xarray surface on top of `oracle/xr_min.py`: what the reference's TESTS call (the package itself needs
only xr_min).

TEST INFRASTRUCTURE -- used by `oracle/run_reference_suite.py` in the build container only.  `extend(xr)` adds to the module
that was loaded under the name `xarray`:

* `xarray.testing.assert_allclose / assert_equal / assert_identical` (dims, values, coordinates; identical also name + attrs);
* DataArray: `roll`, `shift`, `diff`, `mean / max / min / all / any`, `T`, `compute / load / persist`, `chunk` (skips the
  test: dask is absent and the product refuses chunked inputs by design), `to_dataset`, `broadcast_like`, `where`,
  `isnull`, `item`, comparison operators, `__array__`, positional `__getitem__`, `sel` on index coordinates;
* Dataset: `isel`, `rename`, `assign_coords`, `drop_vars`, `reset_coords`, `chunk`, `merge`, `update`, `assign`, `copy`;
* `xarray.broadcast`, `xarray.merge`, `xarray.ones_like / zeros_like`.

Semantics are the documented xarray ones for exactly these uses; a difference from real xarray is not caught (the harness
says "pinned modulo the stand-in").
"""
import sys
import types
from collections import OrderedDict

import numpy as np


def extend(xr):
    import pytest

    DataArray, Dataset = xr.DataArray, xr.Dataset

    # ---- testing ---------------------------------------------------------------------------------------------------------
    def _coords_of(obj):
        return {k: (tuple(v[0]), np.asarray(v[1])) for k, v in obj._coords.items()}

    def _check_coords(a, b, close):
        ca, cb = _coords_of(a), _coords_of(b)
        assert set(ca) == set(cb), f"coordinates differ: {sorted(ca)} vs {sorted(cb)}"
        for k in ca:
            assert ca[k][0] == cb[k][0], f"coordinate {k!r}: dims {ca[k][0]} vs {cb[k][0]}"
            if close and ca[k][1].dtype.kind in "fc" or close and cb[k][1].dtype.kind in "fc":
                np.testing.assert_allclose(ca[k][1], cb[k][1], rtol=1e-5, atol=1e-8, equal_nan=True, err_msg=f"coordinate {k!r}")
            else:
                assert np.array_equal(ca[k][1], cb[k][1], equal_nan=ca[k][1].dtype.kind == "f"), f"coordinate {k!r} differs"

    def _check_arrays(a, b, cmp, close):
        assert type(a) is type(b), f"{type(a)} vs {type(b)}"
        if isinstance(a, Dataset):
            assert set(a._vars) == set(b._vars), f"variables differ: {sorted(a._vars)} vs {sorted(b._vars)}"
            for k in a._vars:
                _check_arrays(a[k], b[k], cmp, close)
            _check_coords(a, b, close)
            return
        assert a.dims == b.dims, f"dims differ: {a.dims} vs {b.dims}"
        assert a.shape == b.shape, f"shapes differ: {a.shape} vs {b.shape}"
        cmp(np.asarray(a.data), np.asarray(b.data))
        _check_coords(a, b, close)

    def assert_allclose(a, b, rtol=1e-05, atol=1e-08, decode_bytes=True):
        _check_arrays(a, b, lambda x, y: np.testing.assert_allclose(x, y, rtol=rtol, atol=atol, equal_nan=True), True)

    def _exact(x, y):
        assert np.array_equal(x, y, equal_nan=x.dtype.kind in "fc" and y.dtype.kind in "fc"), f"values differ:\n{x}\n{y}"

    def assert_equal(a, b):
        _check_arrays(a, b, _exact, False)

    def assert_identical(a, b):
        _check_arrays(a, b, _exact, False)
        if isinstance(a, DataArray):
            assert a.name == b.name, f"names differ: {a.name!r} vs {b.name!r}"
        assert dict(a.attrs) == dict(b.attrs), f"attrs differ: {a.attrs} vs {b.attrs}"

    testing = types.ModuleType("xarray.testing")
    testing.assert_allclose, testing.assert_equal, testing.assert_identical = assert_allclose, assert_equal, assert_identical
    xr.testing = testing
    sys.modules["xarray.testing"] = testing

    # ---- DataArray -------------------------------------------------------------------------------------------------------
    def _needs_dask(*a, **k):
        pytest.skip("needs-dask: chunked arrays (dask is not installable here; xgcm_amd refuses dask-backed inputs by design)")

    def roll(self, shifts=None, roll_coords=False, **kw):
        shifts = dict(shifts or {}, **kw)
        data = self.data
        coords = OrderedDict(self._coords)
        for d, s in shifts.items():
            data = np.roll(data, s, axis=self.get_axis_num(d))
            if roll_coords:
                for k, (cd, cv, ca) in list(coords.items()):
                    if d in cd:
                        coords[k] = (cd, np.roll(cv, s, axis=cd.index(d)), ca)
        return self._new(data, self.dims, coords)

    def shift(self, shifts=None, fill_value=np.nan, **kw):
        shifts = dict(shifts or {}, **kw)
        data = self.data
        for d, s in shifts.items():
            ax = self.get_axis_num(d)
            out = np.full(data.shape, fill_value, dtype=np.result_type(data.dtype, np.asarray(fill_value).dtype) if s else data.dtype)
            src = [slice(None)] * data.ndim
            dst = [slice(None)] * data.ndim
            if s > 0:
                src[ax], dst[ax] = slice(None, -s), slice(s, None)
            elif s < 0:
                src[ax], dst[ax] = slice(-s, None), slice(None, s)
            out[tuple(dst)] = data[tuple(src)]
            data = out
        return self._new(data, self.dims)

    def diff(self, dim, n=1, label="upper"):
        ax = self.get_axis_num(dim)
        sel = slice(1, None) if label == "upper" else slice(None, -1)
        return self.isel({dim: sel})._new_data(np.diff(self.data, n=n, axis=ax))

    def _new_data(self, data):
        return self._new(data, self.dims)

    def _reduce(f, nan_f):
        def method(self, dim=None, skipna=None, keep_attrs=False, **kw):
            dims = list(self.dims) if dim is None else ([dim] if isinstance(dim, str) else list(dim))
            axes = tuple(self.get_axis_num(d) for d in dims)
            g = nan_f if (nan_f is not None and self._skip(skipna)) else f
            out_dims = tuple(d for d in self.dims if d not in dims)
            coords = OrderedDict((k, v) for k, v in self._coords.items() if not (set(v[0]) & set(dims)))
            return self._new(g(self.data, axis=axes), out_dims, coords, attrs=self.attrs if keep_attrs else {})
        return method

    def reduce(self, func, dim=None, **kw):
        dims = list(self.dims) if dim is None else ([dim] if isinstance(dim, str) else list(dim))
        axes = tuple(self.get_axis_num(d) for d in dims)
        out_dims = tuple(d for d in self.dims if d not in dims)
        coords = OrderedDict((k, v) for k, v in self._coords.items() if not (set(v[0]) & set(dims)))
        return self._new(func(self.data, axis=axes, **kw), out_dims, coords, attrs={})

    def to_dataset(self, name=None, **kw):
        name = name or self.name
        if name is None:
            raise ValueError("unable to convert unnamed DataArray to a Dataset without providing an explicit name")
        return Dataset({name: self})

    def broadcast_like(self, other, exclude=None):
        dims = tuple(other.dims) + tuple(d for d in self.dims if d not in other.dims)
        sizes = dict(other.sizes, **self.sizes)
        a = xr._aligned(self, dims)
        return self._new(np.broadcast_to(a, tuple(sizes[d] for d in dims)).copy(), dims)

    def where(self, cond, other=np.nan, drop=False):
        c = cond.data if isinstance(cond, DataArray) else np.asarray(cond)
        if isinstance(cond, DataArray) and cond.dims != self.dims:
            c = xr._aligned(cond, self.dims)
        o = other.data if isinstance(other, DataArray) else other
        return self._new(np.where(c, self.data, o), self.dims)

    def isnull(self):
        return self._new(np.isnan(self.data) if self.data.dtype.kind == "f" else np.zeros(self.shape, bool), self.dims)

    def _cmp(op):
        def method(self, other):
            return self._binary(other, op)
        return method

    def getitem(self, key):
        if isinstance(key, str):
            if key not in self._coords and key in self.dims:
                return DataArray(np.arange(self.sizes[key]), dims=(key,), name=key)
            return self.coords[key]
        if isinstance(key, dict):
            return self.isel(key)
        if not isinstance(key, tuple):
            key = (key,)
        if Ellipsis in key:
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        return self.isel({d: k for d, k in zip(self.dims, key)})

    def setitem(self, key, value):
        if isinstance(key, str):
            self._set_coord(key, value)
            return
        if isinstance(key, dict):
            key = tuple(key.get(d, slice(None)) for d in self.dims)
        self.data[key] = value.data if isinstance(value, DataArray) else value

    def sel(self, indexers=None, method=None, **kw):
        idx = dict(indexers or {}, **kw)
        pos = {}
        for d, v in idx.items():
            index = np.asarray(self._coords[d][1]) if d in self._coords else np.arange(self.sizes[d])
            if isinstance(v, slice):
                lo = 0 if v.start is None else int(np.searchsorted(index, v.start, "left"))
                hi = len(index) if v.stop is None else int(np.searchsorted(index, v.stop, "right"))
                pos[d] = slice(lo, hi)
            elif np.ndim(v) == 0:
                hit = np.nonzero(index == v)[0]
                if not len(hit):
                    raise KeyError(v)
                pos[d] = int(hit[0])
            else:
                pos[d] = np.array([int(np.nonzero(index == x)[0][0]) for x in np.asarray(v)])
        return self.isel(pos)

    def swap_dims(self, dims_dict=None, **kw):
        mp = dict(dims_dict or {}, **kw)
        dims = tuple(mp.get(d, d) for d in self.dims)
        coords = OrderedDict((k, (tuple(mp.get(d, d) for d in cd), cv, ca)) for k, (cd, cv, ca) in self._coords.items())
        return self._new(self.data, dims, coords)

    def reset_coords(self, names=None, drop=False):
        if not drop:
            raise NotImplementedError("reset_coords(drop=False)")
        if names is not None:
            names = [names] if isinstance(names, str) else list(names)
            return self._new(self.data, self.dims, OrderedDict((k, v) for k, v in self._coords.items() if k not in names))
        keep = OrderedDict((k, v) for k, v in self._coords.items() if v[0] == (k,) and k in self.dims)
        return self._new(self.data, self.dims, keep)

    for name, fn in dict(roll=roll, shift=shift, diff=diff, _new_data=_new_data, reduce=reduce, to_dataset=to_dataset,
                         broadcast_like=broadcast_like, where=where, isnull=isnull, sel=sel, swap_dims=swap_dims,
                         reset_coords=reset_coords, chunk=_needs_dask, __getitem__=getitem, __setitem__=setitem,
                         mean=_reduce(np.mean, np.nanmean), max=_reduce(np.max, np.nanmax), min=_reduce(np.min, np.nanmin),
                         std=_reduce(np.std, np.nanstd), prod=_reduce(np.prod, np.nanprod),
                         all=_reduce(np.all, None), any=_reduce(np.any, None),
                         compute=lambda self, **k: self, load=lambda self, **k: self, persist=lambda self, **k: self,
                         item=lambda self: self.data.item(), __array__=lambda self, dtype=None, copy=None: np.asarray(self.data, dtype=dtype),
                         __eq__=_cmp(np.equal), __ne__=_cmp(np.not_equal), __lt__=_cmp(np.less), __le__=_cmp(np.less_equal),
                         __gt__=_cmp(np.greater), __ge__=_cmp(np.greater_equal), __pow__=_cmp(np.power),
                         __abs__=lambda self: self._new(np.abs(self.data), self.dims),
                         __bool__=lambda self: bool(self.data), __float__=lambda self: float(self.data),
                         __invert__=lambda self: self._new(~self.data, self.dims)).items():
        setattr(DataArray, name, fn)
    DataArray.__hash__ = None

    def array_ufunc(self, ufunc, method, *inputs, **kw):  # np.sin(da), np.float64(2) * da
        if method != "__call__":
            return NotImplemented
        first = next(x for x in inputs if isinstance(x, DataArray))
        if sum(isinstance(x, DataArray) for x in inputs) == 2 and len(inputs) == 2:
            a, b = inputs
            return a._binary(b, ufunc)
        raw = [x.data if isinstance(x, DataArray) else x for x in inputs]
        return first._new(ufunc(*raw, **kw), first.dims)

    DataArray.__array_ufunc__ = array_ufunc

    def da_getattr(self, key):  # `da.time`: coordinates (and dims without one) as attributes
        if not key.startswith("_"):
            coords = self.__dict__.get("_coords", {})
            if key in coords or key in self.__dict__.get("dims", ()):
                return self[key]
        raise AttributeError(f"'DataArray' object has no attribute {key!r}")

    DataArray.__getattr__ = da_getattr
    DataArray.T = property(lambda self: self.transpose())
    DataArray.nbytes = property(lambda self: self.data.nbytes)
    DataArray.indexes = property(lambda self: {d: self._coords[d][1] for d in self.dims if d in self._coords})

    # ---- Dataset ---------------------------------------------------------------------------------------------------------
    def _ds_map(ds, f, coords_too=True):
        """apply a DataArray -> DataArray function to every variable (and coordinate) of a dataset"""
        out = Dataset(attrs=ds.attrs)
        for k in ds._coords:
            c = f(ds.coords[k]) if coords_too else ds.coords[k]
            out._coords[k] = (c.dims, np.asarray(c.data), dict(c.attrs))
        for k in ds._vars:
            v = f(ds[k])
            out._vars[k] = (v.dims, np.asarray(v.data), dict(v.attrs))
        return out

    def ds_isel(self, indexers=None, drop=False, **kw):
        idx = dict(indexers or {}, **kw)
        return _ds_map(self, lambda a: a.isel({d: i for d, i in idx.items() if d in a.dims}))

    def ds_rename(self, name_dict=None, **kw):
        mp = dict(name_dict or {}, **kw)
        out = Dataset(attrs=self.attrs)
        for k, (cd, cv, ca) in self._coords.items():
            out._coords[mp.get(k, k)] = (tuple(mp.get(d, d) for d in cd), cv, ca)
        for k, (cd, cv, ca) in self._vars.items():
            out._vars[mp.get(k, k)] = (tuple(mp.get(d, d) for d in cd), cv, ca)
        return out

    def ds_assign_coords(self, coords=None, **kw):
        out = self.copy()
        for k, v in dict(coords or {}, **kw).items():
            out._vars.pop(k, None)
            out._coords[k] = xr._coord_tuple(k, v)
        return out

    def ds_drop_vars(self, names, errors="raise"):
        names = [names] if isinstance(names, str) else list(names)
        out = self.copy()
        for n in names:
            out._vars.pop(n, None)
            out._coords.pop(n, None)
        return out

    def ds_reset_coords(self, names=None, drop=False):
        out = self.copy()
        names = [k for k, v in self._coords.items() if v[0] != (k,)] if names is None else ([names] if isinstance(names, str) else list(names))
        for n in names:
            c = out._coords.pop(n)
            if not drop:
                out._vars[n] = c
        return out

    def ds_set_coords(self, names):
        out = self.copy()
        for n in ([names] if isinstance(names, str) else list(names)):
            out._coords[n] = out._vars.pop(n)
        return out

    def ds_merge(self, other, **kw):
        return xr.merge([self, other])

    def ds_update(self, other):
        for k, v in (other.items() if isinstance(other, dict) else other.data_vars.items()):
            self[k] = v
        return self

    def ds_assign(self, variables=None, **kw):
        out = self.copy()
        for k, v in dict(variables or {}, **kw).items():
            out[k] = v
        return out

    def ds_transpose(self, *dims, **kw):
        return _ds_map(self, lambda a: a.transpose(*[d for d in dims if d in a.dims or d is Ellipsis]) if a.ndim > 1 else a)

    def ds_expand_dims(self, dim=None, axis=0, **kw):
        new = dict(dim if isinstance(dim, dict) else ({} if dim is None else {dim: 1}), **kw)

        def grow(a):
            for d, n in reversed(list(new.items())):
                a = a._new(np.repeat(np.expand_dims(a.data, 0), n, axis=0), (d,) + a.dims)
            return a

        return _ds_map(self, grow, coords_too=False)

    def da_expand_dims(self, dim=None, axis=0, **kw):
        if isinstance(dim, dict) or kw:  # {name: length}: a new leading dim of that length
            out = self
            for d, n in reversed(list(dict(dim or {}, **kw).items())):
                n = len(n) if np.ndim(n) else int(n)
                out = out._new(np.repeat(np.expand_dims(out.data, axis), n, axis=axis), out.dims[:axis] + (d,) + out.dims[axis:])
            return out
        dims = [dim] if isinstance(dim, str) else list(dim)
        out = self
        for d in reversed(dims):
            out = out._new(np.expand_dims(out.data, axis), out.dims[:axis] + (d,) + out.dims[axis:])
        return out

    DataArray.expand_dims = da_expand_dims

    def _ds_binary(op):
        def method(self, other):
            return _ds_map(self, lambda a: getattr(a, op)(other), coords_too=False)
        return method

    for _op in ("__mul__", "__rmul__", "__add__", "__radd__", "__sub__", "__truediv__"):
        setattr(Dataset, _op, _ds_binary(_op))

    def ds_items(self):
        return [(k, self[k]) for k in self._vars]

    for name, fn in dict(isel=ds_isel, rename=ds_rename, assign_coords=ds_assign_coords, drop_vars=ds_drop_vars, drop=ds_drop_vars,
                         reset_coords=ds_reset_coords, set_coords=ds_set_coords, merge=ds_merge, update=ds_update, assign=ds_assign,
                         transpose=ds_transpose, items=ds_items, chunk=_needs_dask, expand_dims=ds_expand_dims,
                         compute=lambda self, **k: self, load=lambda self, **k: self,
                         values=lambda self: [self[k] for k in self._vars],
                         __len__=lambda self: len(self._vars)).items():
        setattr(Dataset, name, fn)
    Dataset.chunks = property(lambda self: {})

    # ---- views: `ds[name].attrs[...] = v` and `obj.coords[name] = data` write through, as in xarray -------------------------
    def _view(owner, table, key):
        dims, values, attrs = table[key]
        sub = OrderedDict((k, v) for k, v in owner._coords.items() if set(v[0]) <= set(dims))
        out = DataArray(values, dims=dims, name=key, _raw_coords=sub)
        out.attrs = attrs  # the SAME dict
        return out

    def coords_getitem(self, key):
        return _view(self._o, self._o._coords, key)

    def coords_setitem(self, key, value):
        o = self._o
        if isinstance(o, DataArray):
            o._set_coord(key, value)
        else:
            if isinstance(value, DataArray):
                for ck, cv in value._coords.items():
                    o._coords.setdefault(ck, cv)
            o._vars.pop(key, None)
            o._coords[key] = xr._coord_tuple(key, value)

    def coords_delitem(self, key):
        del self._o._coords[key]

    xr._Coords.__getitem__, xr._Coords.__setitem__, xr._Coords.__delitem__ = coords_getitem, coords_setitem, coords_delitem
    xr._Coords.update = lambda self, other: [coords_setitem(self, k, v) for k, v in dict(other).items()] and None

    def ds_getitem(self, key):
        if isinstance(key, (list, tuple)):
            out = Dataset(attrs=self.attrs)
            for k in key:
                out[k] = self[k]
            return out
        if key in self._vars:
            return _view(self, self._vars, key)
        if key in self._coords:
            return _view(self, self._coords, key)
        if key in self.sizes:  # a dim without a coordinate
            return DataArray(np.arange(self.sizes[key]), dims=(key,), name=key)
        raise KeyError(key)

    Dataset.__getitem__ = ds_getitem

    def _fresh(table, deep):
        return OrderedDict((k, (d, np.array(v) if deep else v, dict(a))) for k, (d, v, a) in table.items())

    def ds_copy(self, deep=False, data=None):  # xarray: a copy never shares its attrs dicts with the original
        out = Dataset(attrs=dict(self.attrs))
        out._vars, out._coords = _fresh(self._vars, deep), _fresh(self._coords, deep)
        return out

    def da_copy(self, deep=True, data=None):
        d = self.data if data is None else np.asarray(data)
        return self._new(d.copy() if (deep and data is None) else d, self.dims, _fresh(self._coords, deep))

    Dataset.copy, DataArray.copy = ds_copy, da_copy

    # ---- module-level ------------------------------------------------------------------------------------------------------
    def broadcast(*args, exclude=None):
        dims, sizes = [], {}
        for a in args:
            for d, n in a.sizes.items():
                if d not in dims:
                    dims.append(d)
                    sizes[d] = n
        def one(a):
            return a._new(np.broadcast_to(xr._aligned(a, dims), tuple(sizes[d] for d in dims)).copy(), tuple(dims))

        out = []
        for a in args:
            if isinstance(a, Dataset):
                out.append(_ds_map(a, one, coords_too=False))
                continue
            coords = OrderedDict()
            for b in args:
                for k, v in b._coords.items():
                    coords.setdefault(k, v)
            out.append(a._new(np.broadcast_to(xr._aligned(a, dims), tuple(sizes[d] for d in dims)).copy(), tuple(dims), coords))
        return tuple(out)

    def merge(objs, **kw):
        out = Dataset()
        for o in objs:
            if isinstance(o, DataArray):
                o = o.to_dataset()
            out.attrs.update(o.attrs) if not out.attrs else None
            for k, v in o._coords.items():
                out._coords.setdefault(k, v)
            for k, v in o._vars.items():
                out._vars[k] = v
        return out

    xr.broadcast, xr.merge = broadcast, merge
    xr.ones_like = lambda a, dtype=None: a._new(np.ones_like(a.data, dtype=dtype), a.dims)
    xr.zeros_like = lambda a, dtype=None: a._new(np.zeros_like(a.data, dtype=dtype), a.dims)
    xr.full_like = lambda a, v, dtype=None: a._new(np.full_like(a.data, v, dtype=dtype), a.dims)
    xr.set_options = lambda **k: __import__("contextlib").nullcontext()
