"""Golden NetCDF-4 file for xgcm_amd.hdf5, written by REAL h5py / libhdf5.

Run in the build container with the image's Anaconda interpreter (the only one that has h5py):

    /opt/conda/bin/python3.9 oracle/make_golden_netcdf4.py      # h5py 3.3.0 over HDF5 1.10.6

-> tests/golden/netcdf4_state.nc (the file) + tests/golden/netcdf4_state.npz (the arrays that went in, as numpy saw them).
The file is laid out the way the netCDF-4 C library lays a classic-model dataset out in HDF5 (its format specification,
"NetCDF-4 Format": one dimension scale per dimension, `DIMENSION_LIST` references on every variable, `_Netcdf4Dimid`, a
dimension without a coordinate variable as a scale named "This is a netCDF dimension but not a netCDF variable.", NC_CHAR
attributes as fixed-length strings, `_NCProperties` on the root) -- which is also what h5netcdf writes and what
`xarray.open_dataset` reads through either engine.  Nothing of xgcm_amd takes part."""
import os

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NC = os.path.join(ROOT, "tests", "golden", "netcdf4_state.nc")
NPZ = os.path.join(ROOT, "tests", "golden", "netcdf4_state.npz")
NT, NZ, NY, NX = 3, 4, 6, 16


def fixed(s):  # NC_CHAR attribute: a fixed-length string
    return np.bytes_(s)


def main():
    rng = np.random.default_rng(61)
    T = rng.standard_normal((NT, NZ, NY, NX))
    T[0, 1, 2, 3] = -999.0           # a land cell: the _FillValue
    T[2, 3, 5, 15] = -999.0
    T[1, 0, 0, 0] = np.nan           # a NaN that was written as such
    S = (rng.standard_normal((NT, NZ, NY, NX)) * 0.1 + 35.0).astype("f4")
    S[1, 2, 3, 4] = np.float32(1e20)  # missing_value
    eta = (rng.standard_normal((NT, NY, NX)) * 100).astype("i2")
    bnds = np.stack([np.arange(NZ) * 10.0, np.arange(NZ) * 10.0 + 10.0], axis=1)
    coords = {"time": np.arange(NT) * 3600.0, "Z": np.arange(NZ) * 10.0 + 5.0, "YC": np.arange(NY) + 0.5, "XC": np.arange(NX) * 2.0 + 1.0}
    it = np.arange(NT, dtype="i8") * 72
    if os.path.exists(NC):
        os.remove(NC)
    with h5py.File(NC, "w", libver="earliest") as f:
        f.attrs["_NCProperties"] = fixed("version=2,netcdf=4.7.4,hdf5=1.10.6")
        f.attrs["title"] = fixed("xgcm_amd golden NetCDF-4 state file")
        f.attrs["Conventions"] = fixed("CF-1.8")
        scales = {}
        for k, (name, vals) in enumerate(coords.items()):
            d = f.create_dataset(name, data=vals, maxshape=(None,) if name == "time" else None, chunks=(1,) if name == "time" else None)
            d.make_scale(name)
            d.attrs["_Netcdf4Dimid"] = np.int32(k)
            d.attrs["units"] = fixed({"time": "s", "Z": "m", "YC": "degrees_north", "XC": "degrees_east"}[name])
            scales[name] = d
        nv = f.create_dataset("nv", shape=(2,), dtype=">f4")  # a dimension without a coordinate variable
        nv.make_scale("This is a netCDF dimension but not a netCDF variable.         2")
        nv.attrs["_Netcdf4Dimid"] = np.int32(4)
        scales["nv"] = nv

        def var(name, data, dims, **kw):
            v = f.create_dataset(name, data=data, **kw)
            for i, dname in enumerate(dims):
                v.dims[i].attach_scale(scales[dname])
            return v

        v = var("T", T, ("time", "Z", "YC", "XC"), chunks=(1, 2, 3, 8), compression="gzip", compression_opts=4, shuffle=True, fillvalue=-999.0)
        v.attrs["_FillValue"] = np.array([-999.0])
        v.attrs["units"] = fixed("degC")
        v.attrs["long_name"] = "potential temperature"   # a variable-length string (h5netcdf / h5py style)
        v.attrs["coordinates"] = fixed("iter")
        v = var("S", S, ("time", "Z", "YC", "XC"))                        # contiguous, float32
        v.attrs["missing_value"] = np.array([1e20], dtype="f4")
        v.attrs["units"] = fixed("psu")
        v = var("Tbe", T.astype(">f4"), ("time", "Z", "YC", "XC"), chunks=(3, 1, 6, 16), fletcher32=True)  # big-endian on disk
        v.attrs["units"] = fixed("degC")
        v = var("eta", eta, ("time", "YC", "XC"), chunks=(2, 6, 16), compression="gzip")
        v.attrs["valid_range"] = np.array([-500, 500], dtype="i2")
        var("iter", it, ("time",), chunks=(2,), maxshape=(None,))
        var("Z_bnds", bnds, ("Z", "nv"))
        f.create_dataset("rho0", data=np.float64(1029.0))               # a scalar variable
        sp = f.create_dataset("sparse", shape=(NZ, NY, NX), dtype="f8", chunks=(2, 3, 8), fillvalue=-1.0, compression="gzip")
        sp[0:2, 0:3, 0:8] = T[0, 0:2, 0:3, 0:8]                          # one chunk of eight written: the others were never allocated
        for i, dname in enumerate(("Z", "YC", "XC")):
            sp.dims[i].attach_scale(scales[dname])
        v = var("station", np.array([b"alpha", b"beta", b"gamma"], dtype="S8"), ("time",))  # NC_CHAR-like: not a numeric field
        v = var("packed", (eta // 2).astype("i2"), ("time", "YC", "XC"))  # CF packing: refused by name
        v.attrs["scale_factor"] = np.float32(0.01)
        v.attrs["add_offset"] = np.float32(10.0)
    np.savez_compressed(NPZ, T=T, S=S, eta=eta, Z_bnds=bnds, iter=it, **{"c_" + k: v for k, v in coords.items()})
    print(f"{NC}: {os.path.getsize(NC)} bytes; {NPZ}: {os.path.getsize(NPZ)} bytes; h5py {h5py.__version__}, HDF5 {h5py.version.hdf5_version}")


if __name__ == "__main__":
    main()
