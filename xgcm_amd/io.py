"""Block readers for what MITgcm writes, without xarray ("on-disk formats").

The reference never opens a file itself: xarray's backends decode NetCDF / zarr / MDS into (dask-chunked) arrays and
`apply_ufunc(dask="parallelized")` walks the chunks (`xgcm/grid.py:786-818`).  xarray and netCDF4-python are not in this
image, so the formats a model run is kept in are read here directly:

* MDS (`<prefix>.meta` + `<prefix>.data`): one text header, one raw big-endian array `(records, [Nr,] Ny, Nx)`;
* NetCDF-3 classic / 64-bit offset (what `pkg/mnc` writes), through `scipy.io.netcdf_file(mmap=True)`
  -- both as ITERABLES OF RECORD BLOCKS, the shape `xgcm_amd.streaming.stream_blocks / iter_stream` take, straight from a
  memory map of the file;
* zarr format 2 directory stores (what `xarray.Dataset.to_zarr` writes; round 6), as CHUNKED CONTAINERS the operators walk
  block by block (`ZarrArray`, `open_zarr`, `write_zarr`; xgcm_amd.chunked) -- codecs none / blosc (zarr's default) / zstd /
  lz4 / zlib / gzip / bz2 / lzma;
* NetCDF-4 / HDF5 (`open_netcdf4`, `H5Array`, `write_netcdf4`: xgcm_amd.hdf5), chunked containers over libhdf5 hyperslab reads
  and results written back block by block -- where a libhdf5 can be loaded (the image's Anaconda tree has one; the library is never bundled or guessed at).

MDS and NetCDF-3 store big-endian numbers.  Their blocks are handed over AS STORED: `iter_stream` copies the raw bytes into
page-locked memory, sends them over PCIe and reverses the byte order on the GPU (`xg_bswap`), so no host core touches the
values.  `MdsWriter` is the matching sink.  Nothing here computes; there is no CPU path to fall back to.

Tiled MDS output (`<prefix>.001.001.data`, one file per tile: what a run without `globalFiles` / `useSingleCpuIO` leaves) is
assembled by `mds_tiled_blocks`: every tile's records are copied, as stored, into their place in a global block -- a strided
copy on the host, the one pass over the bytes that the staging copy of `iter_stream` would make anyway.

Not covered (say so loudly rather than guess): zarr filters and format 3, blosc's own `blosclz` codec without a libblosc,
packed variables (`scale_factor` / `add_offset`: `netcdf_blocks` and `H5Array` refuse those), HDF5 types other than integers
and floats; tiles that overlap or leave gaps are refused.
"""

from __future__ import annotations

import os
import re
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["read_mds_meta", "mds_blocks", "mds_tile_files", "mds_tiled_blocks", "MdsWriter", "write_mds", "write_mds_tiled",
           "netcdf_blocks", "netcdf_variable_info", "ZarrArray", "open_zarr", "write_zarr", "H5Array", "open_netcdf4", "write_netcdf4"]

from .hdf5 import H5Array, open_netcdf4, write_netcdf4  # noqa: E402,F401 -- NetCDF-4 / HDF5: its own module (ctypes over libhdf5)

_PREC = {"float32": ">f4", "float64": ">f8", "real*4": ">f4", "real*8": ">f8"}


def read_mds_meta(path: str, allow_tile: bool = False) -> Dict:
    """Parse an MDS `.meta` header: {"shape": (records, [Nr,] Ny, Nx) in C order, "dtype": '>f4' | '>f8',
    "nrecords", "dims" (fastest first, as stored), "fields", "timestep"}.

    `dimList` holds one (global extent, first index, last index) triple per dimension, fastest dimension first; a
    triple that does not span its whole extent means a per-tile file, which is refused unless `allow_tile`: then "shape" is
    the TILE's array and "global_dims" / "first" / "last" (fastest first, 1-based inclusive as stored) place it."""
    if not path.endswith(".meta"):
        path = path + ".meta"
    with open(path, "r") as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)  # newer headers carry /* comments */

    def field(name: str) -> Optional[str]:
        m = re.search(name + r"\s*=\s*[\[{](.*?)[\]}]\s*;", text, flags=re.S)
        return m.group(1) if m else None

    ndims = field("nDims")
    dim_list = field("dimList")
    prec = field("dataprec") or field("format")
    if ndims is None or dim_list is None or prec is None:
        raise ValueError(f"{path}: not an MDS header (nDims / dimList / dataprec missing)")
    nd = int(ndims.split()[0])
    nums = [int(v) for v in re.findall(r"-?\d+", dim_list)]
    if len(nums) != 3 * nd:
        raise ValueError(f"{path}: dimList has {len(nums)} numbers for nDims = {nd}")
    dims, gdims, firsts, lasts = [], [], [], []
    for d in range(nd):
        extent, first, last = nums[3 * d: 3 * d + 3]
        if not 1 <= first <= last <= extent:
            raise ValueError(f"{path}: dimension {d} covers {first}..{last} of {extent}")
        if (first != 1 or last != extent) and not allow_tile:
            raise NotImplementedError(f"{path}: a per-tile MDS file (dimension {d} covers {first}..{last} of {extent}); "
                                      "global files are read by mds_blocks, tiled output by mds_tiled_blocks")
        dims.append(last - first + 1)
        gdims.append(extent)
        firsts.append(first)
        lasts.append(last)
    key = prec.replace("'", "").replace('"', "").strip().lower()
    if key not in _PREC:
        raise ValueError(f"{path}: unknown dataprec {prec!r}")
    nrec = int((field("nrecords") or "1").split()[0])
    flds = field("fldList")
    step = field("timeStepNumber")
    shape = (nrec,) + tuple(reversed(dims))
    return {"shape": shape, "dtype": _PREC[key], "nrecords": nrec, "dims": dims, "global_dims": gdims, "first": firsts, "last": lasts,
            "fields": [s.strip() for s in re.findall(r"'([^']*)'", flds)] if flds else [],
            "timestep": int(step.split()[0]) if step and step.split() else None}


def mds_blocks(prefix: str, records_per_block: int = 1, records: Optional[Sequence[int]] = None) -> Iterator[np.ndarray]:
    """Yield the records of `<prefix>.data` in blocks `(n, [Nr,] Ny, Nx)`, big-endian as stored (views of one read-only
    memory map: nothing is read until a block is consumed).  `records = (start, stop)` restricts the range -- a rank's
    shard of the record axis (`xgcm_amd.sharding.shard_bounds`)."""
    if records_per_block < 1:
        raise ValueError("records_per_block must be >= 1")
    base = prefix[:-5] if prefix.endswith((".meta", ".data")) else prefix
    meta = read_mds_meta(base + ".meta")
    shape = meta["shape"]
    want = int(np.prod(shape)) * np.dtype(meta["dtype"]).itemsize
    have = os.path.getsize(base + ".data")
    if have != want:
        raise ValueError(f"{base}.data holds {have} bytes, its header describes {want}")
    lo, hi = (0, shape[0]) if records is None else (int(records[0]), int(records[1]))
    if not 0 <= lo <= hi <= shape[0]:
        raise ValueError(f"records {lo}..{hi} outside 0..{shape[0]}")
    if hi == lo:
        return
    mm = np.memmap(base + ".data", dtype=meta["dtype"], mode="r", shape=shape)
    for s in range(lo, hi, records_per_block):
        yield mm[s: min(s + records_per_block, hi)]


def mds_tile_files(prefix: str) -> List[str]:
    """the per-tile files of `<prefix>`: `<prefix>.<bi>.<bj>` (three digits each, without extension), sorted"""
    import glob

    base = prefix[:-5] if prefix.endswith((".meta", ".data")) else prefix
    hits = sorted(h[:-5] for h in glob.glob(glob.escape(base) + ".[0-9][0-9][0-9].[0-9][0-9][0-9].meta"))
    return hits


def mds_tiled_blocks(prefix: str, records_per_block: int = 1, records: Optional[Sequence[int]] = None) -> Iterator[np.ndarray]:
    """Tiled MDS output as GLOBAL record blocks `(n, [Nr,] Ny, Nx)`, big-endian as stored: every tile file
    `<prefix>.<bi>.<bj>.data` is memory-mapped and its part of a block copied into place (bytes untouched).  The tiles must
    cover the global (Ny, Nx) domain exactly once and agree on precision, record count and the other dimensions."""
    if records_per_block < 1:
        raise ValueError("records_per_block must be >= 1")
    files = mds_tile_files(prefix)
    if not files:
        raise FileNotFoundError(f"no tile files {prefix}.NNN.NNN.meta")
    tiles = []
    ref = None
    for f in files:
        m = read_mds_meta(f + ".meta", allow_tile=True)
        key = (m["dtype"], m["nrecords"], tuple(m["global_dims"]), tuple(m["dims"][2:]))
        if ref is None:
            ref = key
        elif key != ref:
            raise ValueError(f"{f}.meta disagrees with {files[0]}.meta on precision / records / global extents")
        if any(fi != 1 or la != ex for fi, la, ex in zip(m["first"][2:], m["last"][2:], m["global_dims"][2:])):
            raise NotImplementedError(f"{f}.meta: tiled along a dimension other than the two fastest")
        want = int(np.prod(m["shape"])) * np.dtype(m["dtype"]).itemsize
        have = os.path.getsize(f + ".data")
        if have != want:
            raise ValueError(f"{f}.data holds {have} bytes, its header describes {want}")
        tiles.append((m, np.memmap(f + ".data", dtype=m["dtype"], mode="r", shape=m["shape"])))
    m0 = tiles[0][0]
    if len(m0["global_dims"]) < 2:
        raise ValueError(f"{files[0]}.meta: fewer than two dimensions")
    gnx, gny = m0["global_dims"][0], m0["global_dims"][1]
    cover = np.zeros((gny, gnx), dtype=np.uint8)
    for m, _ in tiles:
        cover[m["first"][1] - 1: m["last"][1], m["first"][0] - 1: m["last"][0]] += 1
    if not np.all(cover == 1):
        raise ValueError(f"{prefix}: the tiles do not cover the {gny} x {gnx} domain exactly once "
                         f"({int((cover == 0).sum())} cells missing, {int((cover > 1).sum())} covered twice)")
    nrec = m0["nrecords"]
    gshape = (nrec,) + tuple(reversed(m0["global_dims"]))
    lo, hi = (0, nrec) if records is None else (int(records[0]), int(records[1]))
    if not 0 <= lo <= hi <= nrec:
        raise ValueError(f"records {lo}..{hi} outside 0..{nrec}")
    for s in range(lo, hi, records_per_block):
        e = min(s + records_per_block, hi)
        blk = np.empty((e - s,) + gshape[1:], dtype=m0["dtype"])
        for m, mm in tiles:
            blk[..., m["first"][1] - 1: m["last"][1], m["first"][0] - 1: m["last"][0]] = mm[s:e]
        yield blk


class MdsWriter:
    """Sink for `stream_blocks(fn, blocks, sink=writer.sink)`: appends result blocks to `<prefix>.data` (big-endian) and
    writes the header on `close()`.  Record shape and precision are taken from the first block."""

    def __init__(self, prefix: str, fields: Sequence[str] = (), timestep: int = 0):
        self.prefix = prefix
        self.fields = list(fields)
        self.timestep = int(timestep)
        self._f = open(prefix + ".data", "wb")
        self._rec_shape: Optional[Tuple[int, ...]] = None
        self._dtype: Optional[np.dtype] = None
        self.nrecords = 0

    def sink(self, k: int, block: np.ndarray) -> None:
        block = np.asarray(block)
        if self._rec_shape is None:
            self._rec_shape = tuple(block.shape[1:])
            self._dtype = np.dtype(">f4") if block.dtype == np.float32 else np.dtype(">f8")
        if tuple(block.shape[1:]) != self._rec_shape:
            raise ValueError(f"block {k} has record shape {block.shape[1:]}, the file holds {self._rec_shape}")
        self._f.write(np.ascontiguousarray(block, dtype=self._dtype).tobytes())
        self.nrecords += block.shape[0]

    def close(self) -> None:
        self._f.close()
        if self._rec_shape is None:
            raise ValueError("nothing was written")
        dims = list(reversed(self._rec_shape))
        lines = [f" nDims = [ {len(dims):3d} ];", " dimList = ["]
        lines += [f" {n:5d}, {1:5d}, {n:5d}" + ("," if i + 1 < len(dims) else "") for i, n in enumerate(dims)]
        lines += [" ];", f" dataprec = [ '{'float32' if self._dtype.itemsize == 4 else 'float64'}' ];",
                  f" nrecords = [ {self.nrecords:5d} ];", f" timeStepNumber = [ {self.timestep:10d} ];"]
        if self.fields:
            lines += [f" nFlds = [ {len(self.fields):4d} ];", " fldList = {", " " + " ".join(f"'{n:<8s}'" for n in self.fields), " };"]
        with open(self.prefix + ".meta", "w") as f:
            f.write("\n".join(lines) + "\n")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        else:
            self._f.close()


def write_mds(prefix: str, array: np.ndarray, fields: Sequence[str] = (), timestep: int = 0) -> None:
    """`array` = (records, [Nr,] Ny, Nx) -> `<prefix>.data` / `.meta` the way MITgcm's `mdsio` writes a global file."""
    with MdsWriter(prefix, fields, timestep) as w:
        w.sink(0, np.asarray(array))


def write_mds_tiled(prefix: str, array: np.ndarray, tiles: Tuple[int, int], fields: Sequence[str] = (), timestep: int = 0) -> List[str]:
    """`array` = (records, [Nr,] Ny, Nx) -> one `<prefix>.<bi>.<bj>.data` / `.meta` pair per tile of a `tiles = (nty, ntx)`
    decomposition, the way mdsio writes without `globalFiles` (bi counts along X, bj along Y, from 1).  Returns the prefixes."""
    a = np.asarray(array)
    nty, ntx = tiles
    ny, nx = a.shape[-2], a.shape[-1]
    if ny % nty or nx % ntx:
        raise ValueError(f"{ny} x {nx} does not divide into {nty} x {ntx} tiles")
    ty, tx = ny // nty, nx // ntx
    dt = np.dtype(">f4") if a.dtype == np.float32 else np.dtype(">f8")
    gdims = list(reversed(a.shape[1:]))
    out = []
    for bj in range(nty):
        for bi in range(ntx):
            p = f"{prefix}.{bi + 1:03d}.{bj + 1:03d}"
            sub = a[..., bj * ty: (bj + 1) * ty, bi * tx: (bi + 1) * tx]
            with open(p + ".data", "wb") as f:
                f.write(np.ascontiguousarray(sub, dtype=dt).tobytes())
            rng = [(nx, bi * tx + 1, (bi + 1) * tx), (ny, bj * ty + 1, (bj + 1) * ty)] + [(n, 1, n) for n in gdims[2:]]
            lines = [f" nDims = [ {len(rng):3d} ];", " dimList = ["]
            lines += [f" {n:5d}, {fi:5d}, {la:5d}" + ("," if i + 1 < len(rng) else "") for i, (n, fi, la) in enumerate(rng)]
            lines += [" ];", f" dataprec = [ '{'float32' if dt.itemsize == 4 else 'float64'}' ];",
                      f" nrecords = [ {a.shape[0]:5d} ];", f" timeStepNumber = [ {int(timestep):10d} ];"]
            if fields:
                lines += [f" nFlds = [ {len(fields):4d} ];", " fldList = {", " " + " ".join(f"'{n:<8s}'" for n in fields), " };"]
            with open(p + ".meta", "w") as f:
                f.write("\n".join(lines) + "\n")
            out.append(p)
    return out


def netcdf_variable_info(path: str, name: str) -> Dict:
    """dims, shape, dtype and whether `name` is a record variable of a NetCDF-3 file"""
    from scipy.io import netcdf_file

    with netcdf_file(path, "r", mmap=False) as nc:
        v = nc.variables[name]
        return {"dims": tuple(v.dimensions), "shape": tuple(int(s) for s in v.shape), "dtype": v.data.dtype.str,
                "isrec": bool(v.isrec), "attrs": {k: getattr(v, k) for k in v._attributes}}


def netcdf_missing_value(path: str, name: str) -> Optional[float]:
    """The `_FillValue` (else `missing_value`) of a NetCDF-3 variable, or None.  xarray (mask_and_scale=True, the decoding
    the reference's inputs go through) turns those cells into NaN; hand the value to `iter_stream(..., mask_value=...)` /
    `stream_blocks` and the blocks are masked in HBM after the byte swap (xg_mask_value)."""
    attrs = netcdf_variable_info(path, name)["attrs"]
    for key in ("_FillValue", "missing_value"):
        if key in attrs:
            return float(np.asarray(attrs[key]).reshape(-1)[0])
    return None


def netcdf_blocks(path: str, name: str, records_per_block: int = 1, records: Optional[Sequence[int]] = None,
                  missing: str = "refuse") -> Iterator[np.ndarray]:
    """Yield variable `name` of a NetCDF-3 file in blocks of its FIRST dimension (the record / time axis), big-endian as
    stored, from the file's memory map.  Record variables are interleaved per record on disk: the block is then a strided
    view, which the staging copy of `iter_stream` gathers.  float32 / float64 variables without CF packing only.

    A variable that declares `_FillValue` / `missing_value` (MITgcm's mnc can) is refused unless `missing="raw"`: the raw
    blocks still hold the fill value where the reference's pipeline (xarray decoding) holds NaN, and skipna reductions /
    cumsum over land cells would differ silently.  With `missing="raw"` pass `mask_value=netcdf_missing_value(path, name)`
    to the streaming call and the cells become NaN on the GPU."""
    from scipy.io import netcdf_file
    import warnings

    if records_per_block < 1:
        raise ValueError("records_per_block must be >= 1")
    nc = netcdf_file(path, "r", mmap=True)
    try:
        if name not in nc.variables:
            raise KeyError(f"{path}: no variable {name!r} (has {sorted(nc.variables)})")
        v = nc.variables[name]
        if v.data.dtype.kind != "f" or v.data.dtype.itemsize not in (4, 8):
            raise NotImplementedError(f"{path}:{name} is {v.data.dtype}; float32 / float64 variables only")
        if any(a in v._attributes for a in ("scale_factor", "add_offset")):
            raise NotImplementedError(f"{path}:{name} is a packed variable (scale_factor / add_offset)")
        if missing not in ("refuse", "raw"):
            raise ValueError("missing must be 'refuse' or 'raw'")
        if missing == "refuse" and any(a in v._attributes for a in ("_FillValue", "missing_value")):
            raise NotImplementedError(
                f"{path}:{name} declares _FillValue / missing_value: xarray would mask those cells to NaN.  Read it with "
                "missing='raw' and pass mask_value=xgcm_amd.io.netcdf_missing_value(path, name) to iter_stream / "
                "stream_blocks (masked in HBM), or accept the raw fill values knowingly.")
        if len(v.shape) < 1:
            raise ValueError(f"{path}:{name} is a scalar")
        n = int(v.shape[0])
        lo, hi = (0, n) if records is None else (int(records[0]), int(records[1]))
        if not 0 <= lo <= hi <= n:
            raise ValueError(f"records {lo}..{hi} outside 0..{n}")
        data = v.data
        for s in range(lo, hi, records_per_block):
            yield data[s: min(s + records_per_block, hi)]
        del data, v
    finally:
        with warnings.catch_warnings():  # views handed out may still be alive: the map then outlives the file object
            warnings.simplefilter("ignore", RuntimeWarning)
            nc.close()


# ------------------------------------------------------------------------------------------------------
# zarr (format 2, directory store): what `xarray.Dataset.to_zarr` leaves behind and what the reference's users
# open with `xr.open_zarr` into dask-chunked arrays (`xgcm/grid.py:786-818` walks those chunks).  zarr-python is
# not in this image; the format is a JSON header per array (`.zarray`), a chunk per file (`0.3.1`) and xarray's dim names in
# `.zattrs` (`_ARRAY_DIMENSIONS`), so it is read here directly.  A `ZarrArray` is a CHUNKED CONTAINER in the sense of
# xgcm_amd.chunked (`.chunks`, `.shape`, `.dtype`, slicing): handed to a DataArray, the operators walk its chunks block
# by block -- a chunk file is read (and inflated) when its block is due, never the whole array.
# Codecs: none, zlib, gzip, bz2, lzma (python's own); zstd and lz4 (numcodecs' framings) through the system's libzstd /
# liblz4 (ctypes); blosc -- zarr's DEFAULT compressor, i.e. what an `xarray.Dataset.to_zarr` without an encoding holds --
# decoded here: the c-blosc 1 container (16-byte header, block starts, per-block split streams, byte / bit unshuffle) around
# lz4 / lz4hc / zstd / zlib streams; its own `blosclz` codec only where a libblosc can be loaded.  Pinned against chunks the
# real c-blosc 1.21 / libzstd / liblz4 compressed (tests/golden/codec_chunks.npz; generator: make_golden_codecs.py).
# Filters: refused.
# ------------------------------------------------------------------------------------------------------
_CLIBS: Dict[str, object] = {}


def _clib(name: str):
    """libzstd / liblz4 / libblosc through ctypes, loaded once; None when this box has none"""
    if name not in _CLIBS:
        import ctypes
        import ctypes.util

        lib = None
        cands = [os.environ.get(f"XG_{name.upper()}_LIB"), ctypes.util.find_library(name), f"lib{name}.so.1", f"/opt/conda/lib/lib{name}.so.1"]
        for cand in cands:
            if cand:
                try:
                    lib = ctypes.CDLL(cand)
                    break
                except OSError:
                    continue
        if lib is not None:
            c = ctypes
            if name == "zstd":
                lib.ZSTD_decompress.restype, lib.ZSTD_decompress.argtypes = c.c_size_t, [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]
                lib.ZSTD_getFrameContentSize.restype, lib.ZSTD_getFrameContentSize.argtypes = c.c_ulonglong, [c.c_char_p, c.c_size_t]
                lib.ZSTD_isError.restype, lib.ZSTD_isError.argtypes = c.c_uint, [c.c_size_t]
                lib.ZSTD_createDCtx.restype, lib.ZSTD_createDCtx.argtypes = c.c_void_p, []
                lib.ZSTD_freeDCtx.restype, lib.ZSTD_freeDCtx.argtypes = c.c_size_t, [c.c_void_p]
                lib.ZSTD_decompressDCtx.restype = c.c_size_t
                lib.ZSTD_decompressDCtx.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]
            elif name == "lz4":
                lib.LZ4_decompress_safe.restype, lib.LZ4_decompress_safe.argtypes = c.c_int, [c.c_void_p, c.c_void_p, c.c_int, c.c_int]
            elif name == "blosc":
                lib.blosc_decompress_ctx.restype, lib.blosc_decompress_ctx.argtypes = c.c_int, [c.c_char_p, c.c_void_p, c.c_size_t, c.c_int]
        _CLIBS[name] = lib
    return _CLIBS[name]


def _need(name: str, what: str):
    lib = _clib(name)
    if lib is None:
        raise NotImplementedError(f"{what} needs lib{name} (not found on this box; XG_{name.upper()}_LIB names one)")
    return lib


def _zstd_decode(raw: bytes, nbytes: Optional[int] = None) -> bytes:
    lib = _need("zstd", "zarr compressor 'zstd'")
    if nbytes is None:
        nbytes = lib.ZSTD_getFrameContentSize(raw, len(raw))
        if nbytes >= 2 ** 63:  # ZSTD_CONTENTSIZE_UNKNOWN / _ERROR: numcodecs writes one frame with its size
            raise ValueError("zstd chunk without a content size in its frame header")
    out = np.empty(max(1, int(nbytes)), dtype="u1")
    _inner_decode("zstd", np.frombuffer(raw, dtype="u1").ctypes.data if raw else 0, len(raw), out.ctypes.data, int(nbytes))
    return out[:nbytes].data  # (a buffer, not a copy of one: numpy.frombuffer takes it as it is)


def _lz4_block(raw: bytes, nbytes: int) -> bytes:
    out = np.empty(max(1, nbytes), dtype="u1")
    _inner_decode("lz4", np.frombuffer(raw, dtype="u1").ctypes.data if raw else 0, len(raw), out.ctypes.data, nbytes)
    return out[:nbytes].data


def _inner_decode(fmt: str, src: int, nsrc: int, dst: int, ndst: int, dctx: Optional[int] = None) -> None:
    """one lz4 block / zstd frame / zlib stream at address `src` into `ndst` bytes at address `dst` (no copies on the way);
    `dctx`: a zstd decompression context to reuse (a blosc chunk is thousands of small frames: one context for all of them)"""
    import ctypes

    if fmt == "lz4":
        if _need("lz4", "lz4 streams").LZ4_decompress_safe(ctypes.c_void_p(src), ctypes.c_void_p(dst), nsrc, ndst) != ndst:
            raise ValueError("corrupt lz4 stream in a zarr chunk")
    elif fmt == "zstd":
        lib = _need("zstd", "zstd streams")
        if dctx:
            n = lib.ZSTD_decompressDCtx(ctypes.c_void_p(dctx), ctypes.c_void_p(dst), ndst, ctypes.c_void_p(src), nsrc)
        else:
            n = lib.ZSTD_decompress(ctypes.c_void_p(dst), ndst, ctypes.c_void_p(src), nsrc)
        if lib.ZSTD_isError(n) or n != ndst:
            raise ValueError("corrupt zstd stream in a zarr chunk")
    else:
        import zlib

        got = zlib.decompress(ctypes.string_at(src, nsrc))
        if len(got) != ndst:
            raise ValueError("corrupt zlib stream in a zarr chunk")
        ctypes.memmove(dst, got, ndst)


_BLOSC_FORMATS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}  # header flags bits 5-7 (lz4hc shares lz4's format)


def _blosc_decode(raw: bytes, use_lib: bool = True) -> bytes:
    """One c-blosc 1 buffer -> the chunk's bytes.  Header: version, versionlz, flags (1 byte shuffle, 2 plain copy, 4 bit shuffle,
    16 blocks not split, bits 5-7 the inner format), typesize, then uint32 LE nbytes / blocksize / cbytes; int32 block starts;
    a block is `typesize` streams (one per byte position of the shuffled elements: when split) or one, each an int32 length
    followed by the inner codec's stream -- or by the plain bytes when the length equals the stream's decoded size.

    Where a libblosc loads, it does the work (about 2 GB/s on one core against 0.6 GB/s of the walk below: 345 MB of a smooth
    float64 field, lz4 or zstd + shuffle, page faults of the fresh buffer included); the walk serves every box that has
    liblz4 / libzstd only.  The golden chunks pin both."""
    if len(raw) < 16:
        raise ValueError("blosc chunk shorter than its header")
    flags, typesize = raw[2], raw[3]
    nbytes, blocksize, cbytes = (int.from_bytes(raw[o:o + 4], "little") for o in (4, 8, 12))
    if cbytes != len(raw):
        raise ValueError(f"blosc chunk of {len(raw)} bytes whose header says {cbytes}")
    if flags & 2:  # stored as it was
        return raw[16:16 + nbytes]
    fmt = _BLOSC_FORMATS.get(flags >> 5)
    lib = _clib("blosc") if use_lib or fmt in ("blosclz", "snappy", None) else None
    if lib is not None:
        out = np.empty(max(1, nbytes), dtype="u1")
        if lib.blosc_decompress_ctx(raw, out.ctypes.data, nbytes, 1) != nbytes:
            raise ValueError("corrupt blosc chunk")
        return out[:nbytes].data
    if fmt in ("blosclz", "snappy", None):  # no decoder of ours: the library itself, where there is one
        raise NotImplementedError(f"blosc chunk with inner codec {fmt!r}: served are lz4 / lz4hc / zstd / zlib (libblosc, which "
                                  "has the others, was not found; XG_BLOSC_LIB names one)")
    if nbytes == 0:
        return b""
    if blocksize <= 0 or typesize <= 0:
        raise ValueError("corrupt blosc chunk (header)")
    nblocks = (nbytes + blocksize - 1) // blocksize
    if 16 + 4 * nblocks > len(raw):
        raise ValueError("corrupt blosc chunk (block starts)")
    src = np.frombuffer(raw, dtype="u1")
    starts = np.frombuffer(raw, dtype="<i4", count=nblocks, offset=16)
    out, tmp = np.empty(nbytes, dtype="u1"), np.empty(blocksize, dtype="u1")
    shuffled = (flags & 1 and typesize > 1) or flags & 4
    dctx = _need("zstd", "blosc chunks with zstd inside").ZSTD_createDCtx() if fmt == "zstd" else None
    try:
        _blosc_walk(raw, src, starts, out, tmp, fmt, flags, typesize, nbytes, blocksize, shuffled, dctx)
    finally:
        if dctx:
            _clib("zstd").ZSTD_freeDCtx(dctx)
    return out.data


def _blosc_walk(raw, src, starts, out, tmp, fmt, flags, typesize, nbytes, blocksize, shuffled, dctx) -> None:
    nblocks = len(starts)
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = not (flags & 16) and typesize <= 16 and blocksize // typesize >= 128 and not leftover
        nstreams = typesize if split else 1
        per = bsize // nstreams
        into = tmp if shuffled else out[b * blocksize:b * blocksize + bsize]  # (unshuffled blocks decode in place)
        pos = int(starts[b])
        for k in range(nstreams):
            if pos < 0 or pos + 4 > len(raw):
                raise ValueError("corrupt blosc chunk (stream start)")
            clen = int.from_bytes(raw[pos:pos + 4], "little", signed=True)
            pos += 4
            if clen < 0 or pos + clen > len(raw):
                raise ValueError("corrupt blosc chunk (stream length)")
            if clen == per:
                into[k * per:(k + 1) * per] = src[pos:pos + clen]
            else:
                _inner_decode(fmt, src.ctypes.data + pos, clen, into.ctypes.data + k * per, per, dctx)
            pos += clen
        if not shuffled:
            continue
        dst = out[b * blocksize:b * blocksize + bsize]
        nel = bsize // typesize
        if flags & 1 and typesize > 1:  # byte shuffle: byte j of every element, then byte j + 1 ...; trailing bytes as they are
            dst[:nel * typesize].reshape(nel, typesize)[...] = tmp[:nel * typesize].reshape(typesize, nel).T
            dst[nel * typesize:] = tmp[nel * typesize:bsize]
        elif nel % 8 == 0 and nel:  # bit shuffle (bitshuffle's element transpose): row (j, k) holds bit k of byte j of every
            # element, 8 elements to a byte; c-blosc leaves a block whose element count is no multiple of 8 as it is
            bits = np.unpackbits(tmp[:nel * typesize].reshape(typesize * 8, nel // 8), axis=1, bitorder="little")
            dst[:nel * typesize] = np.packbits(bits.T.reshape(nel, typesize, 8), axis=2, bitorder="little").reshape(-1)
            dst[nel * typesize:] = tmp[nel * typesize:bsize]
        else:
            dst[...] = tmp[:bsize]


def _zarr_decode(raw: bytes, codec: Optional[dict]) -> bytes:
    if codec is None:
        return raw
    cid = codec.get("id")
    if cid == "blosc":
        return _blosc_decode(raw)
    if cid == "zstd":
        return _zstd_decode(raw)
    if cid == "lz4":  # numcodecs.LZ4: the raw size (uint32, little endian), then one lz4 block
        return _lz4_block(raw[4:], int.from_bytes(raw[:4], "little"))
    if cid == "zlib":
        import zlib

        return zlib.decompress(raw)
    if cid == "gzip":
        import gzip

        return gzip.decompress(raw)
    if cid == "bz2":
        import bz2

        return bz2.decompress(raw)
    if cid == "lzma":
        import lzma

        return lzma.decompress(raw)
    raise NotImplementedError(f"zarr compressor {cid!r} is not served (served: none, blosc, zstd, lz4, zlib, gzip, bz2, lzma)")


_ZARR_CODECS = ("blosc", "zstd", "lz4", "zlib", "gzip", "bz2", "lzma")


class ZarrArray:
    """One array of a zarr-2 directory store, read chunk by chunk.  `.chunks` is zarr's chunk SHAPE; `.dims` xarray's
    `_ARRAY_DIMENSIONS` (or None); `x[slices]` (unit-step slices) reads exactly the chunk files the slices cross; a chunk
    that was never written holds the fill value."""

    def __init__(self, path: str, mask: bool = True):
        import json

        self.path = path
        with open(os.path.join(path, ".zarray")) as f:
            meta = json.load(f)
        if meta.get("zarr_format") != 2:
            raise NotImplementedError(f"{path}: zarr format {meta.get('zarr_format')} (format 2 is read here)")
        if meta.get("filters"):
            raise NotImplementedError(f"{path}: zarr filters {[f.get('id') for f in meta['filters']]} are not served")
        self.shape = tuple(int(n) for n in meta["shape"])
        self.chunks = tuple(int(n) for n in meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.ndim = len(self.shape)
        self._order = meta.get("order", "C")
        self._codec = meta.get("compressor")
        self._sep = meta.get("dimension_separator", ".")
        fv = meta.get("fill_value")
        self._fill = (np.nan if fv in ("NaN", None) and self.dtype.kind == "f" else
                      {"Infinity": np.inf, "-Infinity": -np.inf}.get(fv, 0 if fv is None else fv))
        if self._codec is not None and self._codec.get("id") not in _ZARR_CODECS:
            _zarr_decode(b"", self._codec)  # refused at open (by name), not at the first read
        for cid, lib in (("zstd", "zstd"), ("lz4", "lz4")):
            if self._codec is not None and self._codec.get("id") == cid:
                _need(lib, f"zarr compressor {cid!r}")  # ... and so is a library this box lacks
        self.attrs: Dict = {}
        za = os.path.join(path, ".zattrs")
        if os.path.exists(za):
            with open(za) as f:
                self.attrs = json.load(f)
        dims = self.attrs.get("_ARRAY_DIMENSIONS")
        self.dims = tuple(dims) if dims is not None else None
        # CF decoding as xarray's default (`mask_and_scale=True`) applies it to what the reference computes on: cells of a
        # floating-point variable equal to `_FillValue` / `missing_value` are NaN; packed variables are refused by name
        if any(k in self.attrs for k in ("scale_factor", "add_offset")):
            raise NotImplementedError(f"{path} is a packed variable (scale_factor / add_offset)")
        self._missing = []
        if mask and self.dtype.kind == "f":
            for key in ("_FillValue", "missing_value"):
                v = self.attrs.get(key)
                if isinstance(v, (int, float)) and v == v:
                    self._missing.append(self.dtype.type(v))

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def _chunk(self, idx: Tuple[int, ...]) -> np.ndarray:
        name = self._sep.join(str(i) for i in idx) if idx else "0"
        f = os.path.join(self.path, *name.split("/")) if self._sep == "/" else os.path.join(self.path, name)
        if not os.path.exists(f):
            return np.full(self.chunks, self._fill, dtype=self.dtype)
        with open(f, "rb") as fh:
            raw = _zarr_decode(fh.read(), self._codec)
        a = np.frombuffer(raw, dtype=self.dtype)
        if a.size != int(np.prod(self.chunks)):
            raise ValueError(f"{f}: {a.size} cells in a chunk of {self.chunks}")
        for v in self._missing:
            if (a == v).any():
                a = np.where(a == v, self.dtype.type(np.nan), a)
        return a.reshape(self.chunks, order=self._order)  # (edge chunks are stored full-size, padded with the fill value)

    def __getitem__(self, key) -> np.ndarray:
        from .chunked import index_spans

        spans, squeeze = index_spans(key, self.shape, "ZarrArray")
        if squeeze:  # x[i]: that one index, the dim dropped
            part = self[tuple(slice(lo, hi) for lo, hi in spans)]
            return part.reshape([n for d, n in enumerate(part.shape) if d not in squeeze])
        import itertools

        ranges = [range(lo // c, (hi - 1) // c + 1) if hi > lo else range(0) for (lo, hi), c in zip(spans, self.chunks)]
        if all(len(r) == 1 and (lo, hi) == (r[0] * c, r[0] * c + c) for r, (lo, hi), c in zip(ranges, spans, self.chunks)):
            return self._chunk(tuple(r[0] for r in ranges))  # exactly one whole chunk: the decoded chunk itself, no second copy
        out = np.empty([b - a for a, b in spans], dtype=self.dtype)

        def place(idx):  # one chunk file: read, decode, copy its part of the slice (chunks write disjoint parts of `out`)
            blk = self._chunk(idx)
            src, dst = [], []
            for d, i in enumerate(idx):
                a = i * self.chunks[d]
                lo, hi = max(a, spans[d][0]), min(a + self.chunks[d], spans[d][1])
                src.append(slice(lo - a, hi - a))
                dst.append(slice(lo - spans[d][0], hi - spans[d][0]))
            out[tuple(dst)] = blk[tuple(src)]

        from .chunked import pmap

        pmap(place, itertools.product(*ranges))  # the chunk files of ONE request side by side (inflating releases the GIL)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self[(slice(None),) * self.ndim]
        return a if dtype is None else a.astype(dtype)

    def __repr__(self) -> str:
        return f"ZarrArray({self.path!r}, shape={self.shape}, dtype={self.dtype}, chunks={self.chunks})"


def open_zarr(path: str):
    """A zarr-2 GROUP as `xarray.Dataset.to_zarr` writes it -> `xgcm_amd.Dataset`: index coordinates (a variable named like its
    only dim) and the variables `coordinates` attributes name are read now (they are small), every other variable stays a
    `ZarrArray` behind its DataArray -- read chunk by chunk when an operator walks it.  A single array directory passes too."""
    from .labeled import DataArray, Dataset

    if os.path.exists(os.path.join(path, ".zarray")):
        z = ZarrArray(path)
        return DataArray(z, z.dims, name=os.path.basename(os.path.normpath(path)), attrs={k: v for k, v in z.attrs.items() if k != "_ARRAY_DIMENSIONS"})
    names = sorted(n for n in os.listdir(path) if os.path.exists(os.path.join(path, n, ".zarray")))
    if not names:
        raise ValueError(f"{path}: neither a zarr array nor a group of arrays")
    arrays = {}
    for n in names:
        try:
            arrays[n] = ZarrArray(os.path.join(path, n))
        except NotImplementedError as exc:  # a packed variable / an unknown codec next to the fields: left out, by name
            import warnings

            warnings.warn(f"{exc}: variable left out of the dataset", stacklevel=2)
    names = [n for n in names if n in arrays]
    for n, z in arrays.items():
        if z.dims is None:
            raise ValueError(f"{path}/{n}: no `_ARRAY_DIMENSIONS` attribute (a store written by xarray carries the dim names there)")
    listed = {c for z in arrays.values() for c in str(z.attrs.get("coordinates", "")).split()}
    is_coord = {n for n, z in arrays.items() if z.dims == (n,) or n in listed}
    clean = lambda z: {k: v for k, v in z.attrs.items() if k not in ("_ARRAY_DIMENSIONS", "coordinates")}  # noqa: E731
    coords = {n: (arrays[n].dims, np.asarray(arrays[n]), clean(arrays[n])) for n in names if n in is_coord}
    data = {n: DataArray(arrays[n], arrays[n].dims, name=n, attrs=clean(arrays[n])) for n in names if n not in is_coord}
    return Dataset(data, coords)


def _c_compress(name: str, raw: bytes, typesize: int) -> bytes:
    """`raw` as numcodecs' Blosc(lz4, 5, SHUFFLE) / Zstd(1) / LZ4() would store it, through the C library"""
    import ctypes as c

    lib = _need(name, f"zarr compressor {name!r}")
    if name == "blosc":
        lib.blosc_compress_ctx.restype = c.c_int
        lib.blosc_compress_ctx.argtypes = [c.c_int, c.c_int, c.c_size_t, c.c_size_t, c.c_char_p, c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t, c.c_int]
        dst = np.empty(len(raw) + 16 + 4096, dtype="u1")
        n = lib.blosc_compress_ctx(5, 1, typesize, len(raw), raw, dst.ctypes.data, dst.size, b"lz4", 0, 1)
    elif name == "zstd":
        lib.ZSTD_compressBound.restype, lib.ZSTD_compressBound.argtypes = c.c_size_t, [c.c_size_t]
        lib.ZSTD_compress.restype, lib.ZSTD_compress.argtypes = c.c_size_t, [c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t, c.c_int]
        dst = np.empty(lib.ZSTD_compressBound(len(raw)), dtype="u1")
        n = lib.ZSTD_compress(dst.ctypes.data, dst.size, raw, len(raw), 1)
        if lib.ZSTD_isError(n):
            n = -1
    else:
        lib.LZ4_compressBound.restype, lib.LZ4_compressBound.argtypes = c.c_int, [c.c_int]
        lib.LZ4_compress_default.restype, lib.LZ4_compress_default.argtypes = c.c_int, [c.c_char_p, c.c_void_p, c.c_int, c.c_int]
        dst = np.empty(4 + lib.LZ4_compressBound(len(raw)), dtype="u1")
        dst[:4] = np.frombuffer(len(raw).to_bytes(4, "little"), dtype="u1")
        n = lib.LZ4_compress_default(raw, dst.ctypes.data + 4, len(raw), dst.size - 4)
        n = n + 4 if n > 0 else -1
    if n <= 0:
        raise OSError(f"lib{name} could not compress a chunk of {len(raw)} bytes")
    return dst[:n].tobytes()


def write_zarr(path: str, array, chunks: Sequence[int], dims: Optional[Sequence[str]] = None, compressor: Optional[str] = None,
               attrs: Optional[dict] = None) -> None:
    """`array` (numpy, or any chunked container: a result of the block walk is written block by block, never assembled) as a
    zarr-2 array directory with chunk shape `chunks`; `compressor`: None / "zlib" / "gzip" / "bz2" / "lzma", and -- through the
    libraries `_clib` finds, an error naming the missing one otherwise -- "blosc" (zarr's default: lz4, level 5, byte
    shuffle), "zstd", "lz4" (numcodecs' framings).  Chunk files are compressed and written side by side (`chunked.pmap`)."""
    import itertools
    import json

    shape = tuple(int(n) for n in array.shape)
    chunks = tuple(int(c) for c in chunks)
    dtype = np.dtype(array.dtype)
    if dtype.byteorder == "=":
        dtype = dtype.newbyteorder("<" if np.little_endian else ">")
    os.makedirs(path, exist_ok=True)
    enc = {None: lambda b: b, "zlib": lambda b: __import__("zlib").compress(b, 1), "gzip": lambda b: __import__("gzip").compress(b, 1),
           "bz2": lambda b: __import__("bz2").compress(b), "lzma": lambda b: __import__("lzma").compress(b),
           "blosc": lambda b: _c_compress("blosc", b, dtype.itemsize), "zstd": lambda b: _c_compress("zstd", b, 1),
           "lz4": lambda b: _c_compress("lz4", b, 1)}[compressor]
    if compressor in ("blosc", "zstd", "lz4"):
        _need(compressor, f"zarr compressor {compressor!r}")
    codec = {None: None, "blosc": {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}, "zstd": {"id": "zstd", "level": 1},
             "lz4": {"id": "lz4", "acceleration": 1}}.get(compressor, {"id": compressor, **({"level": 1} if compressor in ("zlib", "gzip") else {})})
    with open(os.path.join(path, ".zarray"), "w") as f:
        json.dump({"zarr_format": 2, "shape": list(shape), "chunks": list(chunks), "dtype": dtype.str, "order": "C", "compressor": codec,
                   "fill_value": "NaN" if dtype.kind == "f" else 0, "filters": None}, f)
    meta = dict(attrs or {})
    if dims is not None:
        meta["_ARRAY_DIMENSIONS"] = list(dims)
    with open(os.path.join(path, ".zattrs"), "w") as f:
        json.dump(meta, f)
    def put(idx):  # one chunk file
        sl = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(idx, chunks, shape))
        part = np.asarray(array[sl], dtype=dtype)
        full = part
        if part.shape != chunks:  # an edge chunk is stored at full chunk size
            full = np.full(chunks, np.nan if dtype.kind == "f" else 0, dtype=dtype)
            full[tuple(slice(0, s) for s in part.shape)] = part
        with open(os.path.join(path, ".".join(str(i) for i in idx) if idx else "0"), "wb") as f:
            f.write(enc(np.ascontiguousarray(full).tobytes()))

    from .chunked import pmap

    pmap(put, itertools.product(*[range((n + c - 1) // c) for n, c in zip(shape, chunks)]))
