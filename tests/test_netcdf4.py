"""NetCDF-4 / HDF5 files as chunked inputs (xgcm_amd/hdf5.py).

The reference's users reach such files through `xr.open_dataset(path, chunks=...)`: dask arrays over libhdf5 hyperslab reads,
walked by `apply_ufunc(dask="parallelized")` (`xgcm/grid.py:786-818`).  The file under test was written by REAL h5py 3.3 /
HDF5 1.10.6 in the netCDF-4 library's layout (oracle/make_golden_netcdf4.py -> tests/golden/netcdf4_state.nc + the arrays that
went in, netcdf4_state.npz); it is read here through the same libhdf5 via ctypes.  Skipped on a box without a loadable libhdf5."""
import os
import threading

import numpy as np
import pytest

from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd import hdf5 as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NC, NPZ = os.path.join(GOLD, "netcdf4_state.nc"), os.path.join(GOLD, "netcdf4_state.npz")
needs_hdf5 = pytest.mark.skipif(not H.hdf5_available(), reason="no loadable libhdf5 on this box (XG_HDF5_LIB)")


@pytest.fixture(params=["oracle-double", "host-abi", pytest.param("hip", marks=pytest.mark.gpu)])
def tbackend(request, monkeypatch):
    if request.param == "oracle-double":
        from oracle import fake_device

        fake_device.install(monkeypatch)
    elif request.param == "host-abi":
        import host_abi_device

        host_abi_device.install(monkeypatch)
    return request.param


def _expected():
    z = np.load(NPZ)
    T = z["T"].copy()
    T[T == -999.0] = np.nan          # xarray's mask_and_scale: _FillValue cells are NaN in what the reference computes on
    S = z["S"].copy()
    S[S == np.float32(1e20)] = np.nan
    return z, T, S


@needs_hdf5
def test_the_file_opens_as_the_netcdf_library_laid_it_out():
    z, T, S = _expected()
    with pytest.warns(UserWarning) as rec:  # a string variable and a packed one sit next to the fields: left out, by name
        ds = H.open_netcdf4(NC)
    said = " ".join(str(w.message) for w in rec)
    assert "station" in said and "string" in said and "packed" in said and "scale_factor" in said
    assert sorted(ds.data_vars) == ["S", "T", "Tbe", "Z_bnds", "eta", "rho0", "sparse"] and sorted(ds.coords) == ["XC", "YC", "Z", "iter", "time"]
    assert ds.attrs["Conventions"] == "CF-1.8" and "_NCProperties" not in ds.attrs
    for name in ("time", "Z", "YC", "XC"):
        np.testing.assert_array_equal(ds[name].values, z["c_" + name])
    assert ds["XC"].attrs == {"units": "degrees_east"}      # `_Netcdf4Dimid`, CLASS, NAME, REFERENCE_LIST stay inside
    t = ds["T"]
    assert isinstance(t.data, H.H5Array) and t.dims == ("time", "Z", "YC", "XC") and t.data.layout == "chunked"
    assert t.chunks == ((1, 1, 1), (2, 2), (3, 3), (8, 8)) and t.dtype == np.float64
    assert t.attrs == {"_FillValue": t.attrs["_FillValue"], "units": "degC", "long_name": "potential temperature"}
    assert list(t.coords) == ["XC", "YC", "Z", "iter", "time"]            # `coordinates = "iter"` rides along `time`
    np.testing.assert_array_equal(ds["iter"].values, z["iter"])
    assert np.array_equal(np.asarray(t.data), T, equal_nan=True) and np.isnan(T).sum() == 3
    assert np.array_equal(t.data[1:3, 1:4, 2:5, 3:11], T[1:3, 1:4, 2:5, 3:11], equal_nan=True)
    assert t.data[:, 2:2].shape == (3, 0, 6, 16)
    s = ds["S"]                                                            # contiguous on disk: one block unless asked otherwise
    assert s.data.layout == "contiguous" and s.dtype == np.float32 and s.chunks == ((3,), (4,), (6,), (16,))
    assert np.array_equal(np.asarray(s.data), S, equal_nan=True) and np.isnan(S).sum() == 1
    be = ds["Tbe"]                                                         # big-endian + fletcher32 on disk: native out of libhdf5
    assert be.dtype == np.float32 and be.data.dtype.isnative and np.array_equal(np.asarray(be.data), z["T"].astype("f4"), equal_nan=True)
    assert ds["eta"].dtype == np.int16 and np.array_equal(np.asarray(ds["eta"].data), z["eta"])
    assert ds["Z_bnds"].dims == ("Z", "nv") and "nv" not in ds.coords     # a dimension without a coordinate variable
    assert np.array_equal(np.asarray(ds["Z_bnds"].data), z["Z_bnds"]) and float(ds["rho0"].values) == 1029.0
    raw = H.H5Array(NC, "T", mask=False)
    assert np.array_equal(np.asarray(raw), z["T"], equal_nan=True)
    rec = t.isel(time=1, Z=slice(1, 3))                                   # a record of the file: just the chunks it crosses
    assert rec.dims == ("Z", "YC", "XC") and np.array_equal(np.asarray(rec.values), T[1, 1:3], equal_nan=True)
    assert np.array_equal(t.data[-1, :, 2], T[-1, :, 2], equal_nan=True)
    re = H.open_netcdf4(NC, chunks={"time": -1, "YC": 2})["S"]            # xarray's `chunks=`
    assert re.chunks == ((3,), (4,), (2, 2, 2), (16,))


@needs_hdf5
def test_operators_walk_a_netcdf4_variable_hyperslab_by_hyperslab(tbackend):
    z, T, S = _expected()
    ds = H.open_netcdf4(NC)
    NT, NZ, NY, NX = T.shape
    gds = Dataset({}, {"XC": ("XC", z["c_XC"]), "XG": ("XG", z["c_XC"] - 1.0), "YC": ("YC", z["c_YC"]), "YG": ("YG", z["c_YC"] - 0.5),
                       "Z": ("Z", z["c_Z"]), "Zl": ("Zl", z["c_Z"] - 5.0), "time": ("time", z["c_time"])})
    grid = Grid(gds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, autoparse_metadata=False)
    tcoords = {"iter": ("time", z["iter"]), "time": ("time", z["c_time"]), "XC": ("XC", z["c_XC"]), "YC": ("YC", z["c_YC"]), "Z": ("Z", z["c_Z"])}
    eager = DataArray(T, ("time", "Z", "YC", "XC"), coords=tcoords, name="T")
    reads = []
    orig = H.H5Array.__getitem__
    H.H5Array.__getitem__ = lambda self, key: (reads.append((self.name, tuple((k.start, k.stop) for k in key))), orig(self, key))[1]
    try:
        got = grid.diff(ds["T"], "X")
    finally:
        H.H5Array.__getitem__ = orig
    # blocks of the non-core dims, whole along X: 3 x 2 x 2 hyperslabs, each once (libhdf5 inflates the two chunks each crosses)
    if tbackend == "oracle-double":  # (the double computes on the assembled array: one read of everything)
        assert len(reads) == 1
    else:
        assert len(reads) == len(set(reads)) == 12 and all(r[1][3] == (0, NX) for r in reads)
    want = grid.diff(eager, "X")
    assert got.dims == want.dims and np.array_equal(np.asarray(got.values), np.asarray(want.values), equal_nan=True)
    for call in (lambda v: grid.interp(v, "Y"), lambda v: grid.cumsum(v, "Z"), lambda v: grid.diff(v, "Z"), lambda v: grid.max(v, "X")):
        g, w = call(ds["T"]), call(eager)
        assert g.dims == w.dims and np.array_equal(np.asarray(g.values), np.asarray(w.values), equal_nan=True)
    if tbackend != "oracle-double":
        assert got.chunks == ds["T"].chunks          # the result in the file's own chunking
    s_eager = DataArray(S, ("time", "Z", "YC", "XC"), name="S")  # float32, contiguous (one block), a missing_value cell
    g, w = grid.diff(ds["S"], "Y"), grid.diff(s_eager, "Y")
    assert g.dtype == np.float32 and np.array_equal(np.asarray(g.values), np.asarray(w.values), equal_nan=True)
    e_eager = DataArray(z["eta"], ("time", "YC", "XC"), name="eta")  # int16 on disk: numpy's integer rules
    g, w = grid.diff(ds["eta"], "X"), grid.diff(e_eager, "X")
    assert g.dtype == w.dtype and np.array_equal(np.asarray(g.values), np.asarray(w.values))


@needs_hdf5
def test_raw_chunk_reads_decoded_here_equal_the_librarys_own_pipeline():
    """chunked variables whose filters are deflate / shuffle / fletcher32 are read as RAW chunks and decoded outside the library
    lock (so blocks fetched side by side inflate side by side); everything must equal what `H5Dread` hands back"""
    z, T, S = _expected()
    for name, filters in (("T", (2, 1)), ("Tbe", (3,)), ("eta", (1,)), ("iter", ()), ("S", None), ("sparse", (1,))):
        a = H.H5Array(NC, name, mask=False)
        assert a._filters == filters, (name, a._filters)
        direct = np.asarray(a)
        part = a[(slice(1, 3),) + (slice(None),) * (a.ndim - 1)]
        a._filters = None                                     # the library's pipeline
        slow = np.asarray(a)
        assert direct.dtype == slow.dtype and direct.dtype.isnative
        assert np.array_equal(direct, slow, equal_nan=True) and np.array_equal(part, slow[1:3], equal_nan=True), name
    sp = H.H5Array(NC, "sparse", mask=False)                  # seven of its eight chunks were never allocated: the fill value,
    got = np.asarray(sp)                                      # through H5Dread (the raw path steps aside)
    assert np.array_equal(got[0:2, 0:3, 0:8], z["T"][0, 0:2, 0:3, 0:8], equal_nan=True)
    rest = got.copy()
    rest[0:2, 0:3, 0:8] = -1.0
    assert (rest == -1.0).all()
    assert np.array_equal(sp[0:2, 0:3, 0:8], got[0:2, 0:3, 0:8], equal_nan=True)   # (an allocated chunk alone: the raw path)


@needs_hdf5
def test_results_go_back_into_a_netcdf4_file_block_by_block(tbackend, tmp_path):
    """`write_netcdf4`: a chunked result is written block by block (its blocks shuffled + deflated by helper threads, handed to
    libhdf5 as stored chunks); the file reads back through this package AND through real h5py (the image's Anaconda
    interpreter) as the netCDF-4 layout: chunk shape = block shape, gzip + shuffle, dimension scales attached by name"""
    import json
    import subprocess

    z, T, S = _expected()
    ds = H.open_netcdf4(NC)
    gds = Dataset({}, {"XC": ("XC", z["c_XC"]), "XG": ("XG", z["c_XC"] - 1.0)})
    grid = Grid(gds, coords={"X": {"center": "XC", "left": "XG"}}, padding={"X": "periodic"}, autoparse_metadata=False)
    got = grid.diff(ds["T"], "X")                       # ('time', 'Z', 'YC', 'XG'), blocks (1, 2, 3, 8) as the file's chunks
    got.attrs.update({"units": "degC", "factor": np.float32(0.5)})
    eta = DataArray(z["eta"], ("time", "YC", "XC"), coords={"time": ("time", z["c_time"], {"units": "s"})}, name="eta")
    out = str(tmp_path / "out.nc")
    H.write_netcdf4(out, {"dTdx": got, "eta": eta, "rho0": DataArray(np.float64(1029.0), ())}, attrs={"title": "results"})
    back = H.open_netcdf4(out)
    want = np.asarray(got.values)
    d = back["dTdx"]
    assert d.dims == got.dims and d.attrs["units"] == "degC" and float(d.attrs["factor"][0]) == 0.5 and back.attrs["title"] == "results"
    assert np.array_equal(np.asarray(d.data), want, equal_nan=True)
    assert np.array_equal(np.asarray(back["eta"].data), z["eta"]) and float(back["rho0"].values) == 1029.0
    np.testing.assert_array_equal(back["time"].values, z["c_time"])
    assert back["time"].attrs == {"units": "s"} and "XC" not in back.coords    # eta's XC came without a coordinate: a bare dimension
    np.testing.assert_array_equal(back["Z"].values, z["c_Z"])                  # (the result carried the file's Z along)
    if tbackend != "oracle-double":
        assert d.data.layout == "chunked" and d.data._filters == (2, 1) and d.chunks == got.chunks
    small = str(tmp_path / "small.nc")                # chunks that DIVIDE the blocks once a block is over `chunk_bytes`
    H.write_netcdf4(small, {"dTdx": got}, chunk_bytes=128)
    sb = H.open_netcdf4(small)["dTdx"]
    assert np.array_equal(np.asarray(sb.data), want, equal_nan=True)
    if tbackend != "oracle-double":
        assert sb.data._native_chunk == (1, 1, 1, 8) and sb.data._filters == (2, 1)   # (1, 2, 3, 8) blocks: 384 B -> 64 B chunks
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        return
    code = ("import h5py, json, numpy as np\n"
            f"f = h5py.File({out!r}, 'r')\n"
            "t = f['dTdx']\n"
            "print(json.dumps({'chunks': list(t.chunks or ()), 'compression': t.compression, 'shuffle': bool(t.shuffle),\n"
            "    'scales': [[s.name for s in t.dims[i].values()] for i in range(t.ndim)], 'sum': float(np.nansum(t[...])),\n"
            "    'nan': int(np.isnan(t[...]).sum()), 'zname': f['XC'].attrs['NAME'].decode(), 'eta': str(f['eta'].dtype)}))\n")
    r = subprocess.run([py39, "-W", "ignore", "-c", code], capture_output=True, text=True, timeout=120)
    if r.returncode != 0 and "No module named" in r.stderr:
        return  # that interpreter has no h5py on this box
    assert r.returncode == 0, r.stderr[-2000:]
    seen = json.loads(r.stdout.strip().splitlines()[-1])
    assert seen["scales"] == [["/time"], ["/Z"], ["/YC"], ["/XG"]] and seen["eta"] == "int16"
    assert seen["zname"].startswith("This is a netCDF dimension but not a netCDF variable.")
    assert seen["sum"] == pytest.approx(float(np.nansum(want)), rel=1e-12) and seen["nan"] == int(np.isnan(want).sum())
    if tbackend != "oracle-double":
        assert seen["chunks"] == [1, 2, 3, 8] and seen["compression"] == "gzip" and seen["shuffle"] is True


@needs_hdf5
def test_concurrent_hyperslab_reads_hold_the_library_lock():
    z, T, S = _expected()
    arr = H.open_netcdf4(NC)["T"].data
    out, errs = {}, []

    def work(k):
        try:
            for _ in range(20):
                out[k] = arr[k % 3:k % 3 + 1, :, :, :]
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs and all(np.array_equal(out[k], T[k % 3:k % 3 + 1], equal_nan=True) for k in range(8))


@needs_hdf5
def test_what_is_not_read_says_so(tmp_path):
    with pytest.raises(OSError, match="not an HDF5"):
        H.open_netcdf4(os.path.join(GOLD, "kats.json"))
    with pytest.raises(KeyError, match="no such dataset"):
        H.H5Array(NC, "nope")
    with pytest.raises(IndexError, match="unit-step"):
        H.H5Array(NC, "T")[::2]
    with pytest.raises(IndexError, match="out of bounds"):
        H.H5Array(NC, "T")[3]
    with pytest.raises(NotImplementedError, match="packed variable"):
        H.H5Array(NC, "packed")
    with pytest.raises(NotImplementedError, match="type class string"):
        H.H5Array(NC, "station")


def test_without_a_libhdf5_every_entry_point_names_the_library(monkeypatch):
    monkeypatch.setattr(H, "_LIB", [None])
    assert not H.hdf5_available()
    with pytest.raises(NotImplementedError, match="libhdf5"):
        H.open_netcdf4(NC)
    with pytest.raises(NotImplementedError, match="libhdf5"):
        H.H5Array(NC, "T")
