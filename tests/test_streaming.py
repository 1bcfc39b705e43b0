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
