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
