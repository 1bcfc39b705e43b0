#!/usr/bin/env python3
"""A model run kept on disk, through the operators, back to disk -- block by block.

    python examples/stores_demo.py [netcdf4-file-or-zarr-store] [variable]

With no arguments: the NetCDF-4 file of the test fixtures (tests/golden/netcdf4_state.nc, written by real h5py / HDF5 1.10.6).
What the reference's user writes with xarray + dask,

    ds = xr.open_dataset(path, chunks={"time": 1})          # dask arrays over the file's chunks
    grid = xgcm.Grid(ds, coords=..., padding=...)
    dTdx = grid.diff(ds.T, "X")                             # apply_ufunc(dask="parallelized") walks the chunks
    dTdx.to_dataset(name="dTdx").to_zarr(out)

reads here

    ds = xgcm_amd.io.open_netcdf4(path, chunks={"time": 1})  # H5Arrays over libhdf5 hyperslab reads (open_zarr: ZarrArrays)
    grid = xgcm_amd.Grid(ds, coords=..., padding=...)
    dTdx = grid.diff(ds["T"], "X")                           # blocks through HBM, copies overlapped, chunks decoded by helper threads
    xgcm_amd.io.write_zarr(out, dTdx.data, chunks, dTdx.dims)  # the result's blocks, never assembled

Needs the GPU library (there is no CPU path); prints what it read, the blocks it walked and a check of one block."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from xgcm_amd import Grid  # noqa: E402
from xgcm_amd import io as IO  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "netcdf4_state.nc")
    name = sys.argv[2] if len(sys.argv) > 2 else "T"
    ds = IO.open_zarr(path) if os.path.isdir(path) else IO.open_netcdf4(path)
    var = ds[name]
    print(f"{path}: {name}{var.dims} {var.dtype}, shape {var.shape}, chunks {var.chunks}, held as {type(var.data).__name__}")
    xdim = var.dims[-1]
    # the last dim as a periodic X axis: its own coordinate = cell centres, left faces half a cell before them
    xc = np.asarray(ds[xdim].values, dtype=float)
    left = xdim + "_left"
    from xgcm_amd import Dataset

    gds = Dataset({}, {xdim: (xdim, xc), left: (left, xc - 0.5 * (xc[1] - xc[0]))})
    grid = Grid(gds, coords={"X": {"center": xdim, "left": left}}, padding={"X": "periodic"}, autoparse_metadata=False)
    out = grid.diff(var, "X")
    print(f"grid.diff(..., 'X') -> dims {out.dims}, result held as {type(out.data).__name__} with chunks {out.chunks}")
    first = tuple(slice(0, c[0]) for c in out.chunks[:-1]) + (slice(None),)  # one block of the outer dims, whole along X
    a = np.asarray(var.data[first])
    want = a - np.roll(a, 1, axis=-1)
    same = np.array_equal(np.asarray(out.data[first]), want, equal_nan=True)
    print(f"first block against numpy: {'bit for bit' if same else 'DIFFERS'}")
    store = os.path.join(tempfile.mkdtemp(prefix="xg_demo_"), "d" + name + "dx.zarr")
    IO.write_zarr(store, out.data, [c[0] for c in out.chunks], out.dims, "zlib")
    back = IO.ZarrArray(store)
    print(f"written block by block to {store}: {back.shape} in chunks of {back.chunks}; read back equal: "
          f"{np.array_equal(np.asarray(back), np.asarray(out.values), equal_nan=True)}")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
