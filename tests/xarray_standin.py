"""A duck-typed stand-in for the `xarray` package (xarray is not installable in the build image), used only by
tests/test_xarray_surface.py to execute the xarray-in / xarray-out bridge of xgcm_amd (`labeled.from_xarray`,
`labeled.to_xarray`, `Grid._wrap_in`).  It implements just the attributes the bridge reads -- `.dims`, `.values`,
`.coords`, `.attrs`, `.name`, `.data_vars`, `.chunks` -- on classes NAMED `DataArray` / `Dataset` that live in a
module NAMED `xarray`, which is how the bridge recognises xarray objects without importing xarray."""

import sys
import types
from collections import OrderedDict

import numpy as np


def install(monkeypatch):
    mod = types.ModuleType("xarray")

    class _Coord:
        def __init__(self, dims, values, attrs=None):
            self.dims = tuple(dims)
            self.values = np.asarray(values)
            self.attrs = dict(attrs or {})
            self.dtype = self.values.dtype

    def _coords(spec, default_dim_ok=True):
        out = OrderedDict()
        for name, c in (spec or {}).items():
            if isinstance(c, _Coord):
                out[name] = c
            elif isinstance(c, tuple):
                out[name] = _Coord((c[0],) if isinstance(c[0], str) else c[0], c[1], c[2] if len(c) > 2 else None)
            else:
                out[name] = _Coord((name,), c)
        return out

    class DataArray:
        def __init__(self, data, dims=None, coords=None, name=None, attrs=None, chunks=None):
            self.values = np.asarray(data)
            self.dims = tuple(dims if dims is not None else ())
            self.coords = _coords(coords)
            self.name = name
            self.attrs = dict(attrs or {})
            self.chunks = chunks  # a tuple of per-dim block sizes = dask-backed in real xarray
            self.shape = self.values.shape
            self.dtype = self.values.dtype
            # `.data`: the array behind the variable -- numpy, or (chunked) what a dask array is to the bridge: a container
            # with `.chunks`, slicing and numpy.asarray (dask is not installable: xgcm_amd.chunked.BlockArray plays it)
            self.data = self.values
            if chunks is not None:
                from xgcm_amd.chunked import BlockArray

                self.data = BlockArray.from_array(self.values, chunks)

        def __getitem__(self, key):
            return self.coords[key]

        def isel(self, **indexers):  # (integers only; coordinates of the kept dims)
            index = tuple(indexers.get(d, slice(None)) for d in self.dims)
            dims = tuple(d for d in self.dims if d not in indexers)
            keep = OrderedDict((k, c) for k, c in self.coords.items() if set(c.dims) <= set(dims))
            return DataArray(self.values[index], dims, keep, self.name, self.attrs)

        def transpose(self, *dims):
            return DataArray(self.values.transpose([self.dims.index(d) for d in dims]), dims, self.coords, self.name, self.attrs)

        def chunk(self, spec):
            blocks = tuple((spec.get(d, n),) * (n // spec.get(d, n)) for d, n in zip(self.dims, self.shape))
            return DataArray(self.values, self.dims, self.coords, self.name, self.attrs, chunks=blocks)

    class Dataset:
        def __init__(self, data_vars=None, coords=None, attrs=None):
            self.coords = _coords(coords)
            self.attrs = dict(attrs or {})
            self.data_vars = OrderedDict()
            for name, v in (data_vars or {}).items():
                self.data_vars[name] = v if isinstance(v, DataArray) else DataArray(v[1], v[0], attrs=v[2] if len(v) > 2 else None, name=name)

        def __getitem__(self, key):
            if key in self.data_vars:
                v = self.data_vars[key]
                cs = OrderedDict((k, c) for k, c in self.coords.items() if set(c.dims) <= set(v.dims))
                cs.update(v.coords)
                return DataArray(v.values, v.dims, cs, key, v.attrs, v.chunks)
            return self.coords[key]

    for cls in (DataArray, Dataset):
        cls.__module__ = "xarray.core." + cls.__name__.lower()
    mod.DataArray, mod.Dataset = DataArray, Dataset
    monkeypatch.setitem(sys.modules, "xarray", mod)
    return mod
