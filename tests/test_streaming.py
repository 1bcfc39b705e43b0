"""Host <-> HBM record streaming (xgcm_amd/streaming.py; SURVEY.md §8 f4, first half)."""

import os

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd.streaming import record_blocks, stream_apply, stream_records


def test_record_blocks():
    assert record_blocks(7, 3) == [(0, 3), (3, 6), (6, 7)]
    assert record_blocks(4, 4) == [(0, 4)]
    assert record_blocks(0, 2) == []
    with pytest.raises(ValueError):
        record_blocks(3, 0)


def test_streaming_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        stream_records(lambda x: x, np.zeros((2, 3)))


@pytest.mark.gpu
@pytest.mark.parametrize("register", [True, False])
@pytest.mark.parametrize("block", [1, 2, 5])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_stream_records_matches_oracle(register, block, dtype):
    from xgcm_amd import device as dev

    src = R.synthetic_field((5, 3, 6, 128), 11).astype(dtype)
    got = stream_records(lambda x: dev.stencil1d("diff", x, 3, 1, 0, "periodic"), src, block=block, register=register)
    assert got.dtype == dtype
    np.testing.assert_array_equal(got, R.stencil1d("diff", src, 3, 1, 0, "periodic"))
    # a result with another record shape (integrate along Z) and a caller-provided output
    out = np.empty((5, 6, 128), dtype=dtype)
    res = stream_records(lambda x: dev.reduce1d(x, 1, None, True), src, block=block, out=out, register=register)
    assert res is out
    np.testing.assert_array_equal(out, src.sum(axis=1))
    assert np.array_equal(src, R.synthetic_field((5, 3, 6, 128), 11).astype(dtype))  # the input is untouched


@pytest.mark.gpu
def test_stream_records_large_blocks_locked_piecewise(tmp_path):
    """arrays above the whole-array threshold are page-locked block by block in helper threads, every block's copy cut at
    the page boundary its neighbour's range ends on (`_BlockPinner.pieces`): records whose size is NOT a multiple of the
    page, into a fresh output and into a caller's; a read-only memory map as the source (locked or not, as the driver
    allows); the same array again afterwards (nothing stays locked behind a call)"""
    from xgcm_amd import device as dev
    from xgcm_amd import streaming as S

    old = S._BlockPinner.WHOLE_BELOW
    S._BlockPinner.WHOLE_BELOW = 1 << 20  # 1 MB: the test arrays take the block-wise path
    try:
        nrec, shape = 7, (3, 41, 517)  # 508 728 bytes per record: no block boundary falls on a page boundary
        src = R.synthetic_field((nrec,) + shape, 21)
        want = R.stencil1d("diff", src, 3, 1, 0, "periodic")
        fn = lambda x: dev.stencil1d("diff", x, 3, 1, 0, "periodic")  # noqa: E731
        for block in (1, 2, 3):
            np.testing.assert_array_equal(stream_records(fn, src, block=block), want)
        out = np.full_like(src, 7.0)
        assert stream_records(fn, src, block=2, out=out) is out
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(stream_records(fn, src, block=2), want)  # the same memory locked again: nothing leaked
        path = tmp_path / "records.bin"
        src.tofile(path)
        mm = np.memmap(path, dtype=np.float64, mode="r", shape=src.shape)
        np.testing.assert_array_equal(stream_records(fn, mm, block=2), want)
        ro = np.empty_like(src)
        ro.flags.writeable = False
        with pytest.raises(ValueError, match="read-only"):
            stream_records(fn, src, block=2, out=ro)
        with pytest.raises(ValueError, match="dtype"):
            stream_records(fn, src, block=2, out=np.empty(src.shape, dtype=np.float32))
    finally:
        S._BlockPinner.WHOLE_BELOW = old


@pytest.mark.gpu
def test_stream_records_survives_a_failing_block():
    """an exception raised by `fn` in the middle of the stream reaches the caller, the helper threads and page locks are
    gone, and the next call on the same arrays runs"""
    from xgcm_amd import device as dev
    from xgcm_amd import streaming as S

    old = S._BlockPinner.WHOLE_BELOW
    S._BlockPinner.WHOLE_BELOW = 1 << 20
    try:
        src = R.synthetic_field((6, 4, 64, 512), 22)
        calls = []

        def fn(x):
            calls.append(1)
            if len(calls) == 3:
                raise RuntimeError("block 3 refuses")
            return dev.stencil1d("interp", x, 3, 0, 1, "extend")

        with pytest.raises(RuntimeError, match="block 3 refuses"):
            stream_records(fn, src, block=1)
        calls.clear()
        calls.extend([0] * 10)  # (never 3 again)
        np.testing.assert_array_equal(stream_records(fn, src, block=1), R.stencil1d("interp", src, 3, 0, 1, "extend"))
    finally:
        S._BlockPinner.WHOLE_BELOW = old


@pytest.mark.gpu
def test_stream_apply_through_the_grid():
    nt, nz, ny, nx = 4, 3, 5, 64
    ds = Dataset(coords={"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, padding="periodic", autoparse_metadata=False)
    a = R.synthetic_field((nt, nz, ny, nx), 12)
    da = DataArray(a, ("time", "Z", "Y", "XC"), coords={"time": np.arange(nt)}, name="T")
    res = stream_apply(lambda blk: grid.interp(blk, "X"), da, "time", block=3)
    assert res.dims == ("time", "Z", "Y", "XG") and res.name == "T"
    np.testing.assert_array_equal(res.values, R.stencil1d("interp", a, 3, 1, 0, "periodic"))
    np.testing.assert_array_equal(res.coords["time"].values, np.arange(nt))
    with pytest.raises(ValueError, match="first"):
        stream_apply(lambda blk: blk, DataArray(a, ("Z0", "time", "Y", "XC")), "time")


def test_block_iterator_fails_loudly_without_gpu():
    import torch

    from xgcm_amd.streaming import stream_blocks

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        stream_blocks(lambda x: x, iter([np.zeros((2, 3))]))


@pytest.mark.gpu
def test_stream_blocks_from_memory_mapped_files(tmp_path):
    """any iterable of record blocks, as a dask / zarr reader would hand them over -- here a
    generator that memory-maps one .npy file per block (ragged last block, one Fortran-ordered block)."""
    from xgcm_amd import device as dev
    from xgcm_amd.streaming import iter_stream, stream_blocks

    nz, ny, nx = 3, 6, 128
    lengths = [2, 2, 3, 1, 2]
    full = R.synthetic_field((sum(lengths), nz, ny, nx), 21)
    paths, start = [], 0
    for i, n in enumerate(lengths):
        blk = full[start:start + n]
        if i == 2:
            blk = np.asfortranarray(blk)  # a reader may hand over non-C-contiguous blocks
        np.save(tmp_path / f"rec_{i}.npy", blk)
        paths.append(tmp_path / f"rec_{i}.npy")
        start += n

    def reader():
        for p in paths:
            yield np.load(p, mmap_mode="r")

    diff_x = lambda x: dev.stencil1d("diff", x, 3, 1, 0, "periodic")  # noqa: E731
    got = stream_blocks(diff_x, reader())
    np.testing.assert_array_equal(got, R.stencil1d("diff", full, 3, 1, 0, "periodic"))
    # results arrive in order, one per block, with the block's own record count; sink form writes them out
    seen = []
    assert stream_blocks(lambda x: dev.reduce1d(x, 1, None, True), reader(), sink=lambda k, r: seen.append((k, r.copy()))) is None
    assert [k for k, _ in seen] == list(range(len(lengths))) and [r.shape[0] for _, r in seen] == lengths
    np.testing.assert_array_equal(np.concatenate([r for _, r in seen]), full.sum(axis=1))
    # the generator form, float32 blocks, a single block and an empty iterable
    one = list(iter_stream(diff_x, [full[:1].astype(np.float32)]))
    assert len(one) == 1 and one[0].dtype == np.float32
    np.testing.assert_array_equal(one[0], R.stencil1d("diff", full[:1].astype(np.float32), 3, 1, 0, "periodic"))
    assert list(iter_stream(diff_x, [])) == []


@pytest.mark.gpu
@pytest.mark.parametrize("in_place", [True, False])
def test_iter_stream_with_a_reader_that_refills_one_buffer(in_place):
    """blocks of 8 MB or more are page-locked in place and copied asynchronously from the READER's
    memory; a generator that refills one buffer (`readinto(buf); yield buf`) must not be asked for its next block before
    that copy has finished -- every block, the first included, arrives uncorrupted; the buffer is locked once"""
    from xgcm_amd import device as dev
    from xgcm_amd.streaming import iter_stream

    nz, ny, nx = 4, 512, 640            # 10 MB per block: above the in-place threshold
    nblocks = 6
    buf = np.empty((1, nz, ny, nx))

    def reader():
        for k in range(nblocks):
            buf[...] = R.synthetic_field(buf.shape, 300 + k)     # overwrites the buffer the previous copy read from
            yield buf

    diff_x = lambda x: dev.stencil1d("diff", x, 3, 1, 0, "periodic")  # noqa: E731
    got = list(iter_stream(diff_x, reader(), in_place=in_place))
    assert len(got) == nblocks
    for k, g in enumerate(got):
        np.testing.assert_array_equal(g, R.stencil1d("diff", R.synthetic_field(buf.shape, 300 + k), 3, 1, 0, "periodic"))


@pytest.mark.gpu
def test_iter_stream_keeps_the_dtype_of_integer_and_bool_results():
    """the staging buffers take the RESULT's dtype (they used to be float32 / float64 only: an int64 or
    bool tensor returned by `fn` came back as floats)"""
    import torch

    from xgcm_amd.streaming import iter_stream, stream_blocks

    a = R.synthetic_field((5, 3, 8, 64), 41)
    counts = lambda x: (x > 0).sum(dim=1)          # int64  # noqa: E731
    masks = lambda x: x > 0                         # bool   # noqa: E731
    got = stream_blocks(counts, (a[i:i + 2] for i in range(0, 5, 2)))
    assert got.dtype == np.int64 and np.array_equal(got, (a > 0).sum(axis=1))
    got = np.concatenate(list(iter_stream(masks, (a[i:i + 2] for i in range(0, 5, 2)))))
    assert got.dtype == np.bool_ and np.array_equal(got, a > 0)
    with pytest.raises(TypeError, match="not served"):
        list(iter_stream(lambda x: x.to(torch.bfloat16), [a[:1]]))


@pytest.mark.gpu
def test_stream_blocks_through_the_grid_api():
    from xgcm_amd.streaming import stream_blocks

    nt, nz, ny, nx = 5, 4, 6, 64
    ds = Dataset(coords={"Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0})
    grid = Grid(ds, coords={"Z": {"center": "Z", "left": "Zl"}}, padding="fill", autoparse_metadata=False)
    a = R.synthetic_field((nt, nz, ny, nx), 22)

    def cumsum_z(x):
        return grid.cumsum(DataArray(x, ("time", "Z", "Y", "X")), "Z").data

    got = stream_blocks(cumsum_z, (a[i:i + 2] for i in range(0, nt, 2)))
    np.testing.assert_array_equal(got, R.grid_cumsum(a, 1, "center", "left", "fill"))


@pytest.mark.gpu
def test_large_host_arrays_are_streamed_blockwise(monkeypatch):
    """numpy in -> numpy out above a size threshold goes block-wise through HBM with the copies overlapped
    (device._streamed): same bits as the one-shot path, metrics sliced along the outermost dim when they have it"""
    from xgcm_amd import device as dev

    a = R.synthetic_field((7, 5, 6, 64), 31)
    a[2, 1, 3, 5] = np.nan
    m2 = R.synthetic_metric((1, 1, 6, 64), 32)          # broadcast along the block dim
    m4 = R.synthetic_metric((7, 5, 6, 64), 33)          # has the block dim: sliced per block
    one_shot = {
        "diff": dev.tohost(dev.stencil1d("diff", a, 3, 1, 0, "periodic", 0.0, m4, m2)),
        "interpY": dev.tohost(dev.stencil1d("interp", a, 2, 0, 1, "extend")),
        "cumsum": dev.tohost(dev.cumsum1d(a, 1, 0, 1, 1, 0, "fill", 0.5, False, True, m2, m4)),
        "reduce": dev.tohost(dev.reduce1d(a, 1, m4, True)),
        "mean": dev.tohost(dev.reduce1d(a, 2, m2, "mean_valid")),
    }
    calls = []
    from xgcm_amd import streaming
    real = streaming.stream_records
    monkeypatch.setattr(streaming, "stream_records", lambda *args, **kw: (calls.append(kw.get("block")), real(*args, **kw))[1])
    monkeypatch.setattr(dev, "HOST_STREAM_MIN_BYTES", 1)
    monkeypatch.setattr(dev, "HOST_STREAM_BLOCK_BYTES", 2 * a[0].nbytes)   # blocks of 2 records: 2 + 2 + 2 + 1
    got = {
        "diff": dev.stencil1d("diff", a, 3, 1, 0, "periodic", 0.0, m4, m2),
        "interpY": dev.stencil1d("interp", a, 2, 0, 1, "extend"),
        "cumsum": dev.cumsum1d(a, 1, 0, 1, 1, 0, "fill", 0.5, False, True, m2, m4),
        "reduce": dev.reduce1d(a, 1, m4, True),
        "mean": dev.reduce1d(a, 2, m2, "mean_valid"),
    }
    assert calls == [2] * 5
    for k in one_shot:
        assert isinstance(got[k], np.ndarray), k
        np.testing.assert_array_equal(got[k], one_shot[k], err_msg=k)
    np.testing.assert_array_equal(got["interpY"], R.stencil1d("interp", a, 2, 0, 1, "extend"))
    # the operator's own axis outermost, or a device-resident input: never streamed
    calls.clear()
    dev.stencil1d("diff", a, 0, 1, 0, "periodic")
    dev.stencil1d("diff", dev.asdevice(a), 3, 1, 0, "periodic")
    assert calls == []
    # and through the Grid API: host DataArray in, host DataArray out
    ds = Dataset(coords={"XC": np.arange(64) + 0.5, "XG": np.arange(64) * 1.0})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, padding="periodic", autoparse_metadata=False)
    out = grid.diff(DataArray(a, ("time", "Z", "Y", "XC")), "X")
    assert calls == [2] and isinstance(out.data, np.ndarray)
    np.testing.assert_array_equal(out.values, R.stencil1d("diff", a, 3, 1, 0, "periodic"))


# ---- readers of what MITgcm writes (xgcm_amd/io.py): MDS .meta / .data and NetCDF-3, big-endian on disk ----------
def _mitgcm_like_files(tmp_path, dtype, nrec=5, shape=(3, 6, 128)):
    """an MDS pair and a NetCDF-3 file (record variable + a fixed one) holding the same synthetic records"""
    from scipy.io import netcdf_file

    from xgcm_amd import io as xio

    a = R.synthetic_field((nrec,) + shape, 41).astype(dtype)
    prefix = str(tmp_path / "T.0000000010")
    xio.write_mds(prefix, a, fields=["THETA"], timestep=10)
    path = str(tmp_path / "state.nc")
    with netcdf_file(path, "w", version=2) as nc:  # 64-bit offset, what pkg/mnc writes
        nc.createDimension("T", None)
        for name, n in zip(("Z", "Y", "X"), shape):
            nc.createDimension(name, n)
        v = nc.createVariable("Temp", np.dtype(dtype).newbyteorder(">"), ("T", "Z", "Y", "X"))
        v[:] = a
        e = nc.createVariable("Eta", np.dtype(dtype).newbyteorder(">"), ("T", "Y", "X"))  # a second record variable: records interleave
        e[:] = a[:, 0]
        d = nc.createVariable("Depth", np.dtype(dtype).newbyteorder(">"), ("Y", "X"))
        d[:] = a[0, 0]
    return a, prefix, path


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mds_and_netcdf_readers_hand_over_the_stored_bytes(tmp_path, dtype):
    """no GPU needed: header parsing, block shapes, ragged last block, rank shards, big-endian views of the memory map"""
    from xgcm_amd import io as xio

    a, prefix, path = _mitgcm_like_files(tmp_path, dtype)
    meta = xio.read_mds_meta(prefix)
    assert meta["shape"] == a.shape and meta["dtype"] == (">f4" if dtype == np.float32 else ">f8")
    assert meta["fields"] == ["THETA"] and meta["timestep"] == 10 and meta["dims"] == [128, 6, 3]
    blocks = list(xio.mds_blocks(prefix, 2))
    assert [b.shape[0] for b in blocks] == [2, 2, 1] and all(not b.dtype.isnative for b in blocks)
    np.testing.assert_array_equal(np.concatenate(blocks).astype(dtype), a)
    np.testing.assert_array_equal(np.concatenate(list(xio.mds_blocks(prefix, 3, records=(1, 4)))).astype(dtype), a[1:4])
    assert list(xio.mds_blocks(prefix, 3, records=(2, 2))) == []
    info = xio.netcdf_variable_info(path, "Temp")
    assert info["dims"] == ("T", "Z", "Y", "X") and info["shape"] == a.shape and info["isrec"]
    nb = list(xio.netcdf_blocks(path, "Temp", 2))
    assert [b.shape for b in nb] == [(2,) + a.shape[1:], (2,) + a.shape[1:], (1,) + a.shape[1:]]
    np.testing.assert_array_equal(np.concatenate([np.asarray(b) for b in nb]).astype(dtype), a)
    np.testing.assert_array_equal(np.concatenate([np.asarray(b) for b in xio.netcdf_blocks(path, "Depth", 4)]).astype(dtype), a[0, 0])
    with pytest.raises(KeyError):
        list(xio.netcdf_blocks(path, "Salt"))
    # a per-tile header and a truncated data file are refused
    with open(prefix + ".meta") as f:
        text = f.read()
    tile = str(tmp_path / "tile")
    with open(tile + ".meta", "w") as f:
        f.write(text.replace("  128,     1,   128", "  256,     1,   128"))
    with pytest.raises(NotImplementedError, match="per-tile"):
        xio.read_mds_meta(tile)
    short = str(tmp_path / "short")
    with open(short + ".meta", "w") as f:
        f.write(text)
    with open(short + ".data", "wb") as f:
        f.write(b"\0" * 64)
    with pytest.raises(ValueError, match="holds 64 bytes"):
        list(xio.mds_blocks(short))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tiled_mds_output_is_assembled_into_global_blocks(tmp_path, dtype):
    """one file per tile (`<prefix>.<bi>.<bj>.data`, what a run without globalFiles leaves): headers with partial ranges,
    global blocks assembled from the tiles' memory maps with the stored bytes, ragged last block, rank shards; tiles that
    are missing, overlap or disagree are refused"""
    from xgcm_amd import io as xio

    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 3, 6, 128)).astype(dtype)
    prefix = str(tmp_path / "T.0000000010")
    files = xio.write_mds_tiled(prefix, a, (2, 4), fields=["THETA"], timestep=10)
    assert len(files) == 8 and xio.mds_tile_files(prefix) == sorted(files) and files[0].endswith(".001.001")
    m = xio.read_mds_meta(files[1] + ".meta", allow_tile=True)  # bi = 2, bj = 1
    assert m["shape"] == (5, 3, 3, 32) and m["first"] == [33, 1, 1] and m["last"] == [64, 3, 3] and m["global_dims"] == [128, 6, 3]
    with pytest.raises(NotImplementedError, match="per-tile"):
        xio.read_mds_meta(files[1] + ".meta")
    blocks = list(xio.mds_tiled_blocks(prefix, 2))
    assert [b.shape for b in blocks] == [(2, 3, 6, 128), (2, 3, 6, 128), (1, 3, 6, 128)] and all(not b.dtype.isnative for b in blocks)
    np.testing.assert_array_equal(np.concatenate(blocks).astype(dtype), a)
    np.testing.assert_array_equal(np.concatenate(list(xio.mds_tiled_blocks(prefix, 3, records=(1, 4)))).astype(dtype), a[1:4])
    assert list(xio.mds_tiled_blocks(prefix, 3, records=(2, 2))) == []
    # a 2-D field (no Nr), one tile per row of tiles
    b2 = rng.standard_normal((2, 6, 128)).astype(dtype)
    p2 = str(tmp_path / "Eta.0000000010")
    xio.write_mds_tiled(p2, b2, (3, 1))
    np.testing.assert_array_equal(np.concatenate(list(xio.mds_tiled_blocks(p2, 1))).astype(dtype), b2)
    # refused: no tiles, a missing tile, a tile written twice under another name (overlap), different precision
    with pytest.raises(FileNotFoundError):
        list(xio.mds_tiled_blocks(str(tmp_path / "nothing")))
    import shutil
    for ext in (".meta", ".data"):
        shutil.move(files[3] + ext, str(tmp_path / "gone") + ext)
    with pytest.raises(ValueError, match="96 cells missing"):
        list(xio.mds_tiled_blocks(prefix))
    for ext in (".meta", ".data"):
        shutil.copy(files[0] + ext, prefix + ".009.009" + ext)
    with pytest.raises(ValueError, match="covered twice"):
        list(xio.mds_tiled_blocks(prefix))
    os.remove(prefix + ".009.009.meta")
    for ext in (".meta", ".data"):
        shutil.move(str(tmp_path / "gone") + ext, files[3] + ext)
    with open(files[3] + ".meta") as f:
        text = f.read()
    with open(files[3] + ".meta", "w") as f:
        f.write(text.replace("float32", "floatXX").replace("float64", "float32").replace("floatXX", "float64"))
    with pytest.raises(ValueError, match="disagrees|holds"):
        list(xio.mds_tiled_blocks(prefix))


def test_mds_header_as_mitgcm_writes_it(tmp_path):
    """the layout of a real mdsio header (comments, blank-padded field names, 3 numbers per line)"""
    from xgcm_amd import io as xio

    p = str(tmp_path / "U.0000000072")
    with open(p + ".meta", "w") as f:
        f.write(" nDims = [   3 ];\n dimList = [\n    90,    1,   90,\n    40,    1,   40,\n    15,    1,   15\n ];\n"
                " dataprec = [ 'float32' ];\n nrecords = [     2 ];\n timeStepNumber = [         72 ];\n"
                " timeInterval = [  8.640000000000E+04 ]; /* seconds */\n nFlds = [    1 ];\n fldList = {\n 'UVEL    '\n };\n")
    m = xio.read_mds_meta(p + ".meta")
    assert m["shape"] == (2, 15, 40, 90) and m["dtype"] == ">f4" and m["fields"] == ["UVEL"] and m["timestep"] == 72


def _netcdf_with_fill(tmp_path, dtype, fill=-999.0):
    from scipy.io import netcdf_file

    a = R.synthetic_field((4, 3, 6, 32), 43).astype(dtype)
    a[:, :, 2, 5:9] = fill   # "land"
    a[1, 0, 0, 0] = fill
    path = str(tmp_path / "masked.nc")
    with netcdf_file(path, "w", version=2) as nc:
        nc.createDimension("T", None)
        for name, n in zip(("Z", "Y", "X"), a.shape[1:]):
            nc.createDimension(name, n)
        v = nc.createVariable("Temp", np.dtype(dtype).newbyteorder(">"), ("T", "Z", "Y", "X"))
        v[:] = a
        v._FillValue = np.array(fill, dtype=np.dtype(dtype).newbyteorder(">"))
        m = nc.createVariable("Salt", np.dtype(dtype).newbyteorder(">"), ("T", "Z", "Y", "X"))
        m[:] = a
        m.missing_value = np.array(fill, dtype=np.dtype(dtype).newbyteorder(">"))
    return a, path


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_netcdf_fill_values_are_not_ignored_silently(tmp_path, dtype):
    """a variable with _FillValue / missing_value holds NaN in the reference's pipeline (xarray decoding); the
    raw blocks are refused unless asked for, and the value to mask is exposed"""
    from xgcm_amd import io as xio

    a, path = _netcdf_with_fill(tmp_path, dtype)
    for name in ("Temp", "Salt"):
        with pytest.raises(NotImplementedError, match="_FillValue / missing_value"):
            list(xio.netcdf_blocks(path, name, 2))
        assert xio.netcdf_missing_value(path, name) == -999.0
        raw = np.concatenate([np.asarray(b) for b in xio.netcdf_blocks(path, name, 2, missing="raw")]).astype(dtype)
        np.testing.assert_array_equal(raw, a)
    with pytest.raises(ValueError):
        list(xio.netcdf_blocks(path, "Temp", 2, missing="whatever"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fill_values_become_nan_in_hbm(tmp_path, dtype):
    """the streamed cumsum / sum of a variable with missing cells equals the reference pipeline's: mask to NaN, then the
    NaN-skipping operators"""
    from xgcm_amd import device as dev
    from xgcm_amd import io as xio
    from xgcm_amd.streaming import stream_blocks

    a, path = _netcdf_with_fill(tmp_path, dtype)
    masked = np.where(a == dtype(-999.0), np.nan, a).astype(dtype)
    mv = xio.netcdf_missing_value(path, "Temp")
    got = stream_blocks(lambda x: dev.cumsum1d(x, 1, 0, 0, 0, 0, None, 0.0, False, True),
                        xio.netcdf_blocks(path, "Temp", 3, missing="raw"), mask_value=mv)
    np.testing.assert_array_equal(got, R.cumsum1d(masked, 1, 0, 0, 0, 0, None, 0.0, False, True))
    ident = stream_blocks(lambda x: dev.stencil1d("max", x, 3, 0, 0, None), xio.netcdf_blocks(path, "Salt", 2, missing="raw"),
                          mask_value=mv)
    assert np.isnan(ident).sum() > 0 and not (ident == dtype(-999.0)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_operators_streamed_over_mds_and_netcdf_files(tmp_path, dtype):
    """diff / cumsum streamed over both on-disk formats, bit-equal to the in-memory operator; the
    big-endian blocks cross PCIe raw and are byte-swapped in HBM (xg_bswap); results written back as MDS."""
    from xgcm_amd import device as dev
    from xgcm_amd import io as xio
    from xgcm_amd.streaming import stream_blocks, stream_records

    a, prefix, path = _mitgcm_like_files(tmp_path, dtype)
    want_diff = R.stencil1d("diff", a, 3, 1, 0, "periodic")
    want_cum = R.cumsum1d(a, 1, 0, 1, 1, 0, "fill", dtype(0), False, True)
    ops = ((lambda x: dev.stencil1d("diff", x, 3, 1, 0, "periodic"), want_diff),
           (lambda x: dev.cumsum1d(x, 1, 0, 1, 1, 0, "fill", 0.0, False, True), want_cum))
    tiled = str(tmp_path / "Ttiled.0000000010")
    xio.write_mds_tiled(tiled, a, (2, 4), fields=["THETA"], timestep=10)  # 8 per-tile files of 3 x 32 cells
    for fn, want in ops:
        for blocks in (xio.mds_blocks(prefix, 2), xio.netcdf_blocks(path, "Temp", 2), xio.netcdf_blocks(path, "Temp", 5),
                       xio.mds_tiled_blocks(tiled, 2)):
            got = stream_blocks(fn, blocks)
            assert got.dtype == dtype
            np.testing.assert_array_equal(got, want)
    # the whole memory-mapped file as ONE host array through the record pipeline
    mm = np.memmap(prefix + ".data", dtype=xio.read_mds_meta(prefix)["dtype"], mode="r", shape=a.shape)
    np.testing.assert_array_equal(stream_records(ops[0][0], mm, block=2), want_diff)
    # results back to disk in the same format, read again
    with xio.MdsWriter(str(tmp_path / "dTdx.0000000010"), fields=["dTdx"], timestep=10) as w:
        stream_blocks(ops[0][0], xio.mds_blocks(prefix, 2), sink=w.sink)
    back = np.concatenate(list(xio.mds_blocks(str(tmp_path / "dTdx.0000000010"), 4))).astype(dtype)
    np.testing.assert_array_equal(back, want_diff)
    # the swap kernel itself: odd element counts (the 16-byte tail), both widths
    import torch
    from xgcm_amd import _hip
    for n in (1, 3, 4, 5, 1027):
        raw = R.synthetic_field((n,), 5).astype(dtype)
        t = torch.from_numpy(raw.view(np.dtype(dtype).newbyteorder(">")).astype(dtype)).cuda()  # bytes reversed on the host
        _hip.check(_hip.load().xg_bswap(t.data_ptr(), n, raw.itemsize, torch.cuda.current_stream().cuda_stream))
        np.testing.assert_array_equal(t.cpu().numpy(), raw)
