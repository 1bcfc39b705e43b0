"""Operator chains captured as ONE hipGraph (xgcm_amd/graphs.py): same bits as the eager calls, replay after replay,
with new input values written into the same storage -- including the chained scan, whose hand-off workspace and ticket
counters must come back clean from every launch for a replay to be valid."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_captured_operator_chain_equals_eager(dtype):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import refimpl as R
    from xgcm_amd import DataArray, Dataset, Grid
    from xgcm_amd import device as D
    from xgcm_amd.graphs import capture

    tdt = getattr(torch, dtype)
    nz, ny, nx = 4, 320, 256
    coords = {"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0, "YC": np.arange(ny) + 0.5, "YG": np.arange(ny) * 1.0,
              "Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0}
    met = lambda shape, seed: D.synthetic(shape, seed, 0, 1.0, 1.0, dtype=tdt)  # noqa: E731
    dv = {"dxC": DataArray(met((ny, nx), 31), ("YC", "XG")), "dyC": DataArray(met((ny, nx), 32), ("YG", "XC")),
          "drF": DataArray(met((nz,), 33), ("Z",)), "rA": DataArray(met((ny, nx), 34), ("YC", "XC"))}
    grid = Grid(Dataset(dv, coords), coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, metrics={("X",): ["dxC"], ("Y",): ["dyC"], ("Z",): ["drF"], ("X", "Y"): ["rA"]},
                autoparse_metadata=False)
    buf = D.synthetic((nz, ny, nx), 2, dtype=tdt)
    T = DataArray(buf, ("Z", "YC", "XC"), name="T")

    def chain():
        return (grid.derivative(T, "X"), grid.interp(T, "Y"), grid.cumsum(T, "Y"),  # cumsum along Y: the chained scan (320 rows)
                grid.integrate(T, "Z"), grid.interp(T, ["X", "Y"]), grid.average(T, ["X", "Y"]), grid.cumint(T, "Y"))

    step = capture(chain)
    for seed in (2, 7, 11):  # new values in the same storage, one graph launch each
        buf.copy_(D.synthetic((nz, ny, nx), seed, dtype=tdt))
        got = [o.data.clone() for o in step()]
        torch.cuda.synchronize()
        want = [o.data for o in chain()]
        for g, w in zip(got, want):
            assert g.dtype == tdt and torch.equal(g, w)
    # and against the oracle for one of them (the capture is not comparing the library with itself only)
    a = D.tohost(buf)
    np.testing.assert_array_equal(D.tohost(step()[2].data), R.cumsum1d(a, 1, 0, 1, 1, 0, "extend", a.dtype.type(0), False, True))
    with pytest.raises(ValueError, match="HBM-resident"):
        capture(lambda: np.zeros(3))
