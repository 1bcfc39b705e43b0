"""Chunked (dask-style) inputs: SURVEY section 8 row f4, VERDICT r05 "next round" 5.

The reference walks the chunks of a dask-backed field (`xgcm/grid.py:786-818`; a chunked core dim: `map_overlap`,
`xgcm/grid_ufunc.py:1057-1133`, refused for inner / outer positions `:1136-1159`).  dask is not installable here: the inputs
are `xgcm_amd.chunked.BlockArray`s (`.chunks`, `.shape`, `.dtype`, slicing) -- the protocol a dask array offers -- and every
result must equal the eager result of the assembled array, on the oracle double, the host build of the C ABI and HIP; on the
two real libraries the result must also come back in the input's chunking, never concatenated."""
import numpy as np
import pytest

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


def _pair(a, chunks):
    dims = ("time", "Z", "YC", "XC")
    tcoord = {"time": ("time", np.arange(NT) * 10.0)}
    return DataArray(a, dims, coords=tcoord, name="T"), DataArray(BlockArray.from_array(a, chunks), dims, coords=tcoord, name="T")


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
def test_two_point_operators_walk_the_blocks(tbackend, op):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS)
    for axis, out_chunks in (("X", CHUNKS), ("Y", CHUNKS), ("Z", CHUNKS)):
        _same(getattr(grid, op)(chunked, axis), getattr(grid, op)(eager, axis), tbackend, out_chunks)
    _same(getattr(grid, op)(chunked, "Z", to="outer"), getattr(grid, op)(eager, "Z", to="outer"), tbackend,
          (CHUNKS[0], (NZ + 1,), CHUNKS[2], CHUNKS[3]))
    _same(getattr(grid, op)(chunked, ["X", "Y"]), getattr(grid, op)(eager, ["X", "Y"]), tbackend, CHUNKS)  # one axis after the other


def test_metrics_ride_block_by_block(tbackend):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS)
    for axis in ("X", "Y", "Z"):
        _same(grid.derivative(chunked, axis), grid.derivative(eager, axis), tbackend, CHUNKS)
        _same(grid.interp(chunked, axis, metric_weighted=(axis,)), grid.interp(eager, axis, metric_weighted=(axis,)), tbackend, CHUNKS)
        _same(grid.cumint(chunked, axis), grid.cumint(eager, axis), tbackend)
    _same(chunked * ds["dyF"], eager * ds["dyF"], tbackend, CHUNKS)
    _same(ds["drF"] * chunked, ds["drF"] * eager, tbackend)


def test_scans_and_sums(tbackend):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS)
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


def test_a_chunked_core_dim_is_read_whole_and_keeps_its_chunks(tbackend):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS_X)
    _same(grid.diff(chunked, "X"), grid.diff(eager, "X"), tbackend, CHUNKS_X)      # center -> left: same length, same chunks
    _same(grid.cumsum(chunked, "X"), grid.cumsum(eager, "X"), tbackend, CHUNKS_X)
    _same(grid.integrate(chunked, "X"), grid.integrate(eager, "X"), tbackend, CHUNKS_X[:3])


def test_inner_outer_along_a_chunked_core_dim_is_the_references_error(tbackend):
    grid, ds, a = _setup()
    zsplit = ((5,), (2, 2), (6,), (16,))
    eager, chunked = _pair(a, zsplit)
    with pytest.raises(NotImplementedError, match="Cannot chunk along a core dimension"):
        grid.diff(chunked, "Z", to="outer")
    _same(grid.diff(chunked, "Z"), grid.diff(eager, "Z"), tbackend, zsplit)                          # center -> left is fine
    _same(grid.cumsum(chunked, "Z", to="outer"), grid.cumsum(eager, "Z", to="outer"), tbackend)    # and cumsum is exempt there too


def test_integer_and_float32_blocks(tbackend):
    grid, ds, a = _setup()
    for arr in ((a * 100).astype(np.float32), np.nan_to_num(a * 1000).astype(np.int32)):
        eager, chunked = _pair(arr, CHUNKS)
        for axis in ("X", "Z"):
            _same(grid.diff(chunked, axis), grid.diff(eager, axis), tbackend, CHUNKS)
        _same(grid.cumsum(chunked, "Z"), grid.cumsum(eager, "Z"), tbackend, CHUNKS)


def test_what_the_chunked_path_does_not_serve_says_so(tbackend):
    grid, ds, a = _setup()
    eager, chunked = _pair(a, CHUNKS)
    with pytest.raises(NotImplementedError, match="does not take dask-chunked inputs"):
        grid.diff({"X": chunked}, "X", other_component={"Y": chunked})


# ----------------------------------------------------------------------------------------------
# zarr-2 directory stores (xarray's `to_zarr` layout) as chunked inputs: read chunk file by chunk file
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("compressor", [None, "zlib", "lzma"])
def test_zarr_store_walked_chunk_by_chunk(tbackend, tmp_path, compressor):
    from xgcm_amd import io as IO

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
    (tmp_path / "a" / "1.1").unlink()  # a chunk that was never written: the fill value
    assert np.isnan(z[3:6, 2:4]).all() and np.array_equal(z[0:3], a[0:3])
    meta = json.loads((tmp_path / "a" / ".zarray").read_text())
    meta["compressor"] = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1}
    (tmp_path / "a" / ".zarray").write_text(json.dumps(meta))
    with pytest.raises(NotImplementedError, match="blosc"):
        IO.ZarrArray(str(tmp_path / "a"))
    meta["compressor"], meta["filters"] = None, [{"id": "delta"}]
    (tmp_path / "a" / ".zarray").write_text(json.dumps(meta))
    with pytest.raises(NotImplementedError, match="filters"):
        IO.ZarrArray(str(tmp_path / "a"))
