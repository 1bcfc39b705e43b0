"""Host <-> HBM record streaming (xgcm_amd/streaming.py; SURVEY.md §8 f4, first half)."""

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
