"""Chunked (dask-style) inputs: SURVEY section 8 row f4, VERDICT r05 "next round" 5.

The reference walks the chunks of a dask-backed field (`xgcm/grid.py:786-818`; a chunked core dim: `map_overlap`,
`xgcm/grid_ufunc.py:1057-1133`, refused for inner / outer positions `:1136-1159`).  dask is not installable here: the inputs
are `xgcm_amd.chunked.BlockArray`s (`.chunks`, `.shape`, `.dtype`, slicing) -- the protocol a dask array offers -- and every
result must equal the eager result of the assembled array, on the oracle double, the host build of the C ABI and HIP; on the
two real libraries the result must also come back in the input's chunking, never concatenated."""
import numpy as np
import pytest

import real_dask
from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd.chunked import BlockArray, block_slices, normalize_chunks


@pytest.fixture(params=["oracle-double", "host-abi", pytest.param("hip", marks=pytest.mark.gpu)])
def tbackend(request, monkeypatch):
    if request.param == "oracle-double":
        from oracle import fake_device

        fake_device.install(monkeypatch)
    elif request.param == "host-abi":
        import host_abi_device

        host_abi_device.install(monkeypatch)
    return request.param


NT, NZ, NY, NX = 5, 4, 6, 16
CHUNKS = ((2, 2, 1), (4,), (3, 3), (16,))          # records and Y split, Z and X whole
CHUNKS_X = ((5,), (4,), (6,), (8, 4, 4))           # the contiguous dim itself split


def _setup():
    coords = {"XC": ("XC", np.arange(NX) + 0.5), "XG": ("XG", np.arange(NX) * 1.0), "YC": ("YC", np.arange(NY) + 0.5),
              "YG": ("YG", np.arange(NY) * 1.0), "Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0),
              "Zp1": ("Zp1", np.arange(NZ + 1) * 1.0), "time": ("time", np.arange(NT) * 10.0)}
    ds = Dataset({"dxC": (("YC", "XG"), R.synthetic_metric((NY, NX), 31)), "dxF": (("YC", "XC"), R.synthetic_metric((NY, NX), 35)),
                  "dyC": (("YG", "XC"), R.synthetic_metric((NY, NX), 32)), "dyF": (("YC", "XC"), R.synthetic_metric((NY, NX), 36)),
                  "drF": (("Z",), R.synthetic_metric((NZ,), 33)), "drC": (("Zl",), R.synthetic_metric((NZ,), 34))}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl", "outer": "Zp1"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"},
                metrics={("X",): ["dxC", "dxF"], ("Y",): ["dyC", "dyF"], ("Z",): ["drF", "drC"]}, autoparse_metadata=False)
    a = R.synthetic_field((NT, NZ, NY, NX), 7)
    a[1, 2, 3, 4] = np.nan
    return grid, ds, a


@pytest.fixture(params=["blockarray", "dask"])
def container(request):
    """what holds the chunked input: this package's own BlockArray, or a REAL dask array (tests/real_dask.py finds one in the
    image's Anaconda tree; skipped on a box without it)"""
    if request.param == "dask" and real_dask.dask_array() is None:
        pytest.skip("no dask on this box (tests/real_dask.py)")
    return request.param


def _pair(a, chunks, container="blockarray"):
    dims = ("time", "Z", "YC", "XC")
    tcoord = {"time": ("time", np.arange(NT) * 10.0)}
    held = BlockArray.from_array(a, chunks) if container == "blockarray" else real_dask.dask_array().from_array(a, chunks=chunks)
    return DataArray(a, dims, coords=tcoord, name="T"), DataArray(held, dims, coords=tcoord, name="T")


def _same(got, want, tbackend, chunks_like=None):
    assert got.dims == want.dims and got.name == want.name and list(got.coords) == list(want.coords)
    assert np.array_equal(np.asarray(got.values), np.asarray(want.values), equal_nan=True)
    if tbackend != "oracle-double":  # (the double computes on the assembled array: values only)
        assert got.chunks is not None and tuple(sum(c) for c in got.chunks) == got.shape
        if chunks_like is not None:
            assert got.chunks == chunks_like, (got.chunks, chunks_like)


def test_block_array_is_the_protocol():
    a = np.arange(5 * 6 * 4.0).reshape(5, 6, 4)
    b = BlockArray.from_array(a, ((2, 3), (1, 5), (4,)))
    assert b.shape == a.shape and b.numblocks == (2, 2, 1) and b.chunks == ((2, 3), (1, 5), (4,))
    assert np.array_equal(np.asarray(b), a) and np.array_equal(b[1:4, 0:3], a[1:4, 0:3]) and b[:, 2:2].shape == (5, 0, 4)
    assert normalize_chunks((2, 4, 4), (5, 6, 4)) == ((2, 2, 1), (4, 2), (4,))  # zarr's chunk shape
    assert [idx for idx, _ in block_slices(b.chunks, whole=(1,))] == [(0, 0, 0), (1, 0, 0)]


@pytest.mark.parametrize("op", ["diff", "interp", "min", "max"])
def test_two_point_operators_walk_the_blocks(tbackend, op, container):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS, container)
    for axis, out_chunks in (("X", CHUNKS), ("Y", CHUNKS), ("Z", CHUNKS)):
        _same(getattr(grid, op)(chunked, axis), getattr(grid, op)(eager, axis), tbackend, out_chunks)
    _same(getattr(grid, op)(chunked, "Z", to="outer"), getattr(grid, op)(eager, "Z", to="outer"), tbackend,
          (CHUNKS[0], (NZ + 1,), CHUNKS[2], CHUNKS[3]))
    _same(getattr(grid, op)(chunked, ["X", "Y"]), getattr(grid, op)(eager, ["X", "Y"]), tbackend, CHUNKS)  # one axis after the other


def test_metrics_ride_block_by_block(tbackend, container):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS, container)
    for axis in ("X", "Y", "Z"):
        _same(grid.derivative(chunked, axis), grid.derivative(eager, axis), tbackend, CHUNKS)
        _same(grid.interp(chunked, axis, metric_weighted=(axis,)), grid.interp(eager, axis, metric_weighted=(axis,)), tbackend, CHUNKS)
        _same(grid.cumint(chunked, axis), grid.cumint(eager, axis), tbackend)
    _same(chunked * ds["dyF"], eager * ds["dyF"], tbackend, CHUNKS)
    _same(ds["drF"] * chunked, ds["drF"] * eager, tbackend)


def test_scans_and_sums(tbackend, container):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS, container)
    _same(grid.cumsum(chunked, "Z"), grid.cumsum(eager, "Z"), tbackend, CHUNKS)
    _same(grid.cumsum(chunked, "Z", to="outer"), grid.cumsum(eager, "Z", to="outer"), tbackend)
    _same(grid.cumsum(chunked, "Y", to="left"), grid.cumsum(eager, "Y", to="left"), tbackend, CHUNKS)  # Y is split: scanned whole
    got, want = grid.integrate(chunked, "Z"), grid.integrate(eager, "Z")
    _same(got, want, tbackend, (CHUNKS[0], CHUNKS[2], CHUNKS[3]))
    _same(grid.integrate(chunked, ["Z", "Y"]), grid.integrate(eager, ["Z", "Y"]), tbackend)
    got, want = grid.average(chunked, "Z"), grid.average(eager, "Z")
    assert got.dims == want.dims and np.allclose(np.asarray(got.values), np.asarray(want.values), rtol=1e-13, equal_nan=True)
    got, want = grid.average(chunked, ["Z", "Y"]), grid.average(eager, ["Z", "Y"])
    assert got.dims == want.dims and np.allclose(np.asarray(got.values), np.asarray(want.values), rtol=1e-12, equal_nan=True)


def test_a_chunked_core_dim_is_read_whole_and_keeps_its_chunks(tbackend, container):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS_X, container)
    _same(grid.diff(chunked, "X"), grid.diff(eager, "X"), tbackend, CHUNKS_X)      # center -> left: same length, same chunks
    _same(grid.cumsum(chunked, "X"), grid.cumsum(eager, "X"), tbackend, CHUNKS_X)
    _same(grid.integrate(chunked, "X"), grid.integrate(eager, "X"), tbackend, CHUNKS_X[:3])


def test_inner_outer_along_a_chunked_core_dim_is_the_references_error(tbackend, container):
    grid, ds, a = _setup()
    zsplit = ((5,), (2, 2), (6,), (16,))
    eager, chunked = _pair(a, zsplit, container)
    with pytest.raises(NotImplementedError, match="Cannot chunk along a core dimension"):
        grid.diff(chunked, "Z", to="outer")
    _same(grid.diff(chunked, "Z"), grid.diff(eager, "Z"), tbackend, zsplit)                          # center -> left is fine
    _same(grid.cumsum(chunked, "Z", to="outer"), grid.cumsum(eager, "Z", to="outer"), tbackend)    # and cumsum is exempt there too


def test_integer_and_float32_blocks(tbackend, container):
    grid, ds, a = _setup()
    for arr in ((a * 100).astype(np.float32), np.nan_to_num(a * 1000).astype(np.int32)):
        eager, chunked = _pair(arr, CHUNKS, container)
        for axis in ("X", "Z"):
            _same(grid.diff(chunked, axis), grid.diff(eager, axis), tbackend, CHUNKS)
        _same(grid.cumsum(chunked, "Z"), grid.cumsum(eager, "Z"), tbackend, CHUNKS)


def test_what_the_chunked_path_does_not_serve_says_so(tbackend, container):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS, container)
    with pytest.raises(NotImplementedError, match="does not take dask-chunked inputs"):
        grid.diff({"X": chunked}, "X", other_component={"Y": chunked})


# ----------------------------------------------------------------------------------------------
# zarr-2 directory stores (xarray's `to_zarr` layout) as chunked inputs: read chunk file by chunk file
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("compressor", [None, "zlib", "lzma", "blosc", "zstd", "lz4"])
def test_zarr_store_walked_chunk_by_chunk(tbackend, tmp_path, compressor):
    from xgcm_amd import io as IO

    if compressor in ("blosc", "zstd", "lz4") and IO._clib(compressor) is None:
        pytest.skip(f"no lib{compressor} on this box")

    grid, ds, a = _setup()
    store = tmp_path / "run.zarr"
    (store).mkdir()
    (store / ".zgroup").write_text('{"zarr_format": 2}')
    dims = ("time", "Z", "YC", "XC")
    IO.write_zarr(str(store / "T"), a, (2, 4, 4, 16), dims, compressor, attrs={"units": "degC", "coordinates": "iter"})
    IO.write_zarr(str(store / "time"), np.arange(NT) * 10.0, (NT,), ("time",), None)
    IO.write_zarr(str(store / "iter"), np.arange(NT, dtype=np.int64) * 72, (2,), ("time",), compressor)
    zds = IO.open_zarr(str(store))
    T = zds["T"]
    assert isinstance(T.data, IO.ZarrArray) and T.dims == dims and T.attrs == {"units": "degC"}
    assert T.chunks == ((2, 2, 1), (4,), (4, 2), (16,)) and list(T.coords) == ["iter", "time"]
    np.testing.assert_array_equal(zds["time"].values, np.arange(NT) * 10.0)
    eager = DataArray(a, dims, coords={"iter": ("time", np.arange(NT, dtype=np.int64) * 72), "time": ("time", np.arange(NT) * 10.0)}, name="T")
    reads = []
    orig = IO.ZarrArray._chunk
    IO.ZarrArray._chunk = lambda self, idx: (reads.append(idx), orig(self, idx))[1]
    try:
        got = grid.diff(T, "X")
    finally:
        IO.ZarrArray._chunk = orig
    assert len(reads) == len(set(reads)) == 6  # every chunk file of T read exactly once: 3 x 1 x 2 x 1
    want = grid.diff(eager, "X")
    assert got.dims == want.dims and np.array_equal(np.asarray(got.values), np.asarray(want.values), equal_nan=True)
    for call in (lambda v: grid.cumsum(v, "Z"), lambda v: grid.derivative(v, "Y"), lambda v: grid.integrate(v, "Z")):
        g, w = call(T), call(eager)
        assert g.dims == w.dims and np.array_equal(np.asarray(g.values), np.asarray(w.values), equal_nan=True)
    if tbackend != "oracle-double":  # the chunked result goes back to a store block by block
        IO.write_zarr(str(store / "dTdx"), got.data, (2, 4, 4, 16), got.dims, compressor)
        back = IO.ZarrArray(str(store / "dTdx"))
        assert np.array_equal(np.asarray(back), np.asarray(want.values), equal_nan=True)


def test_zarr_slices_missing_chunks_and_refusals(tmp_path):
    import json

    from xgcm_amd import io as IO

    a = np.arange(7 * 5, dtype=">f4").reshape(7, 5)
    IO.write_zarr(str(tmp_path / "a"), a, (3, 2), ("y", "x"), "gzip")
    z = IO.ZarrArray(str(tmp_path / "a"))
    assert z.shape == (7, 5) and z.dtype == np.dtype(">f4") and normalize_chunks(z.chunks, z.shape) == ((3, 3, 1), (2, 2, 1))
    assert np.array_equal(z[2:7, 1:4], a[2:7, 1:4]) and np.array_equal(np.asarray(z), a) and z[3:3].shape == (0, 5)
    assert np.array_equal(z[4], a[4]) and np.array_equal(z[-1, 1:3], a[-1, 1:3]) and z[2, 3] == a[2, 3]  # ints drop their dim
    (tmp_path / "a" / "1.1").unlink()  # a chunk that was never written: the fill value
    assert np.isnan(z[3:6, 2:4]).all() and np.array_equal(z[0:3], a[0:3])
    meta = json.loads((tmp_path / "a" / ".zarray").read_text())
    import json as _json

    IO.write_zarr(str(tmp_path / "m"), np.array([[1.5, -999.0, 3.0], [np.nan, 5.0, -999.0]]), (1, 3), ("y", "x"), "zlib", attrs={"_FillValue": -999.0})
    m = IO.ZarrArray(str(tmp_path / "m"))  # xarray's mask_and_scale: `_FillValue` cells of a float variable are NaN
    assert np.array_equal(np.asarray(m), [[1.5, np.nan, 3.0], [np.nan, 5.0, np.nan]], equal_nan=True)
    assert np.array_equal(np.asarray(IO.ZarrArray(str(tmp_path / "m"), mask=False)), [[1.5, -999.0, 3.0], [np.nan, 5.0, -999.0]], equal_nan=True)
    za = _json.loads((tmp_path / "m" / ".zattrs").read_text())
    za["scale_factor"] = 0.01
    (tmp_path / "m" / ".zattrs").write_text(_json.dumps(za))
    with pytest.raises(NotImplementedError, match="packed variable"):
        IO.ZarrArray(str(tmp_path / "m"))
    meta["compressor"] = {"id": "pcodec", "level": 8}
    (tmp_path / "a" / ".zarray").write_text(json.dumps(meta))
    with pytest.raises(NotImplementedError, match="pcodec"):
        IO.ZarrArray(str(tmp_path / "a"))
    meta["compressor"], meta["filters"] = None, [{"id": "delta"}]
    (tmp_path / "a" / ".zarray").write_text(json.dumps(meta))
    with pytest.raises(NotImplementedError, match="filters"):
        IO.ZarrArray(str(tmp_path / "a"))


# ----------------------------------------------------------------------------------------------
# real dask: what is computed, when, and in which form the result leaves
# ----------------------------------------------------------------------------------------------
class _CountingSource:
    """the array behind `dask.array.from_array`: records every block dask fetches"""

    def __init__(self, a):
        self.a, self.shape, self.dtype, self.ndim, self.reads = a, a.shape, a.dtype, a.ndim, []

    def __getitem__(self, key):
        self.reads.append(tuple((k.start, k.stop) for k in key))
        return self.a[key]


def test_real_dask_input_is_computed_chunk_by_chunk_once_and_leaves_as_dask(tbackend):
    """reference: `apply_ufunc(dask="parallelized")` (xgcm/grid.py:786-818) -- nothing is computed when the DataArray is built,
    every chunk is fetched exactly once by the walk, and the result's blocks go back into ONE dask array of the input's chunking"""
    dsa = real_dask.dask_array()
    if dsa is None:
        pytest.skip("no dask on this box (tests/real_dask.py)")
    grid, ds, a = _setup()
    src = _CountingSource(a)
    held = dsa.from_array(src, chunks=CHUNKS, asarray=False, fancy=False, meta=np.empty((0,) * 4, a.dtype))
    dims, tcoord = ("time", "Z", "YC", "XC"), {"time": ("time", np.arange(NT) * 10.0)}
    chunked, eager = DataArray(held, dims, coords=tcoord, name="T"), DataArray(a, dims, coords=tcoord, name="T")
    assert src.reads == [] and chunked.chunks == CHUNKS  # building the labelled array computed nothing
    got = grid.derivative(chunked, "X")
    assert len(src.reads) == len(set(src.reads)) == 3 * 1 * 2 * 1  # each chunk once
    want = grid.derivative(eager, "X")
    assert np.array_equal(np.asarray(got.values), np.asarray(want.values), equal_nan=True)
    if tbackend != "oracle-double":
        out = got.data.to_dask()
        assert isinstance(out, dsa.Array) and out.chunks == CHUNKS and out.dtype == a.dtype
        assert np.array_equal(out.compute(), np.asarray(want.values), equal_nan=True)
        # ... and a dask expression over the input (not yet computed) is walked the same way
        del src.reads[:]
        lazy_in = DataArray((held * 2.0).rechunk({2: NY}), dims, coords=tcoord, name="T")
        got2 = grid.cumsum(lazy_in, "Z")
        assert len(src.reads) == 6  # the six source chunks, once each, through the graph
        assert np.array_equal(np.asarray(got2.values), np.asarray(grid.cumsum(eager * 2.0, "Z").values), equal_nan=True)


# ----------------------------------------------------------------------------------------------
# zarr's default compressor (blosc) and numcodecs' zstd / lz4: decoded here, pinned against chunks the REAL c-blosc 1.21 /
# libzstd / liblz4 compressed (tests/golden/codec_chunks.npz <- oracle/make_golden_codecs.py)
# ----------------------------------------------------------------------------------------------
def _codec_cases():
    import json
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "codec_chunks.npz"))
    return z, json.loads(z["cases"].tobytes())["cases"]


def test_blosc_zstd_lz4_chunks_of_the_real_libraries_decode_bit_for_bit():
    from xgcm_amd import io as IO

    z, cases = _codec_cases()
    assert len(cases) == 70
    seen = set()
    for c in cases:
        comp, raw = c["compressor"], np.ascontiguousarray(z["raw__" + c["field"]]).tobytes()
        if comp["id"] == "blosc" and comp["cname"] == "blosclz" and IO._clib("blosc") is None and not z[c["key"]][2] & 2:
            with pytest.raises(NotImplementedError, match="blosclz"):  # blosc's own codec: only where a libblosc is loadable
                IO._zarr_decode(z[c["key"]].tobytes(), comp)
            continue
        assert IO._zarr_decode(z[c["key"]].tobytes(), comp) == raw, c["key"]   # (through libblosc where one loads)
        if comp["id"] == "blosc" and (comp["cname"] != "blosclz" or z[c["key"]][2] & 2):
            assert IO._blosc_decode(z[c["key"]].tobytes(), use_lib=False) == raw, c["key"]  # this module's own walk of the container
        if comp["id"] == "blosc":
            b = z[c["key"]]
            seen.add(("copy" if b[2] & 2 else "nosplit" if b[2] & 16 else "split", "bit" if b[2] & 4 else "byte" if b[2] & 1 else "none",
                      int.from_bytes(b[4:8].tobytes(), "little") > int.from_bytes(b[8:12].tobytes(), "little")))
    # the container's every branch was walked: plain copies, split and unsplit blocks, the three shuffles, several blocks
    assert {("copy", "byte", False), ("split", "byte", False), ("split", "byte", True), ("split", "none", False),
            ("nosplit", "bit", True), ("nosplit", "byte", True)} <= seen, seen
    for bad in (b"", b"\x02\x01\x21\x08" + b"\x00" * 11):
        with pytest.raises(ValueError):
            IO._blosc_decode(bad)


@pytest.mark.parametrize("which", ["blosc_lz4_5_1_0", "blosc_zstd_1_2_0", "zstd_7", "lz4"])
def test_a_store_as_xarray_writes_it_by_default_is_walked(tbackend, tmp_path, which):
    """`ds.to_zarr(path)` without an encoding: Blosc(lz4, clevel 5, shuffle) chunks.  The store is put together from the real
    library's compressed bytes (one chunk = the whole golden field), opened and differenced block by block"""
    import json

    from xgcm_amd import io as IO

    z, cases = _codec_cases()
    case = next(c for c in cases if c["key"] == "f8_smooth__" + which)
    a = z["raw__f8_smooth"]                                   # (4, 30, 60)
    d = tmp_path / "s.zarr" / "T"
    d.mkdir(parents=True)
    (tmp_path / "s.zarr" / ".zgroup").write_text('{"zarr_format": 2}')
    (d / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": [8, 30, 60], "chunks": [4, 30, 60], "dtype": "<f8", "order": "C",
                                           "compressor": case["compressor"], "fill_value": "NaN", "filters": None}))
    (d / ".zattrs").write_text(json.dumps({"_ARRAY_DIMENSIONS": ["Z", "YC", "XC"]}))
    for name in ("0.0.0", "1.0.0"):                           # two chunks along Z, the same bytes
        (d / name).write_bytes(z[case["key"]].tobytes())
    T = IO.open_zarr(str(tmp_path / "s.zarr"))["T"]
    assert isinstance(T.data, IO.ZarrArray) and T.chunks == ((4, 4), (30,), (60,))
    full = np.concatenate([a, a], axis=0)
    assert np.array_equal(np.asarray(T.data), full)
    coords = {"XC": ("XC", np.arange(60) + 0.5), "XG": ("XG", np.arange(60) * 1.0), "YC": ("YC", np.arange(30) + 0.5), "Z": ("Z", np.arange(8) + 0.5)}
    grid = Grid(Dataset({}, coords), coords={"X": {"center": "XC", "left": "XG"}}, padding={"X": "periodic"}, autoparse_metadata=False)
    got, want = grid.diff(T, "X"), grid.diff(DataArray(full, ("Z", "YC", "XC")), "X")
    assert got.dims == want.dims and np.array_equal(np.asarray(got.values), np.asarray(want.values))


def test_blocks_are_fetched_ahead_in_order_and_errors_surface():
    """`chunked.read_ahead`: helper threads fetch the next blocks while the caller works; order kept, never more than `workers`
    blocks ahead, a failing read raises at its place in the sequence"""
    import threading
    import time

    from xgcm_amd.chunked import read_ahead

    class Slow:
        chunks, shape, dtype = ((1,) * 12,), (12,), np.dtype("f8")

        def __init__(self):
            self.lock, self.active, self.peak, self.threads, self.done = threading.Lock(), 0, 0, set(), []

        def __getitem__(self, sl):
            with self.lock:
                self.active += 1
                self.peak = max(self.peak, self.active)
                self.threads.add(threading.get_ident())
            time.sleep(0.02)
            if sl[0].start == 9 and getattr(self, "fail", False):
                raise OSError("chunk 9 is damaged")
            with self.lock:
                self.active -= 1
                self.done.append(sl[0].start)
            return np.full(1, float(sl[0].start))

    src = Slow()
    sls = [(slice(i, i + 1),) for i in range(12)]
    seen = []
    for blk in read_ahead(src, sls, workers=4):
        seen.append(float(blk[0]))
        assert len(src.done) - len(seen) <= 4          # never further ahead than the helpers there are
    assert seen == [float(i) for i in range(12)] and src.peak > 1 and threading.get_ident() not in src.threads
    src2 = Slow()
    assert [float(b[0]) for b in read_ahead(src2, sls, workers=0)] == seen and src2.threads == {threading.get_ident()}
    bad = Slow()
    bad.fail = True
    got = []
    with pytest.raises(OSError, match="chunk 9"):
        for blk in read_ahead(bad, sls, workers=3):
            got.append(float(blk[0]))
    assert got == [float(i) for i in range(9)]


def test_random_indices_into_every_container_equal_numpy(tmp_path):
    """BlockArray, ZarrArray (edge chunks, blosc where a library loads), H5Array (raw-chunk path and H5Dread) and a real dask array
    under 300 random keys of ints and unit-step slices (negative, empty, out of order) against numpy's indexing of the same array"""
    from xgcm_amd import hdf5 as H
    from xgcm_amd import io as IO

    rng = np.random.default_rng(2026)
    a = rng.standard_normal((7, 5, 11, 13))
    held = {"block": BlockArray.from_array(a, ((3, 3, 1), (5,), (4, 4, 3), (6, 7)))}
    IO.write_zarr(str(tmp_path / "z"), a, (3, 2, 4, 5), None, "blosc" if IO._clib("blosc") is not None else "zlib")
    held["zarr"] = IO.ZarrArray(str(tmp_path / "z"))
    if H.hdf5_available():
        try:
            H.write_netcdf4(str(tmp_path / "a.nc"), {"a": DataArray(held["block"], ("t", "z", "y", "x"))}, chunk_bytes=400)
            held["hdf5"] = H.H5Array(str(tmp_path / "a.nc"), "a")
            assert held["hdf5"]._filters == (2, 1)
            slow = H.H5Array(str(tmp_path / "a.nc"), "a")
            slow._filters = None
            held["hdf5-H5Dread"] = slow
        except NotImplementedError:  # no libhdf5_hl next to libhdf5 on this box: nothing to read back
            pass
    dsa = real_dask.dask_array()
    if dsa is not None:
        held["dask"] = dsa.from_array(a, chunks=(2, 5, 4, 13))
    for trial in range(300):
        key = []
        for n in a.shape[:int(rng.integers(1, 5))]:
            kind = rng.integers(0, 4)
            if kind == 0:
                key.append(int(rng.integers(-n, n)))
            elif kind == 1:
                key.append(slice(None))
            else:
                lo, hi = int(rng.integers(-n - 1, n + 2)), int(rng.integers(-n - 1, n + 2))
                key.append(slice(lo, hi) if kind == 2 else slice(lo, None))
        key = tuple(key)
        want = a[key]
        for label, arr in held.items():
            got = np.asarray(arr[key])
            assert got.shape == want.shape and np.array_equal(got, want), (label, key)
