"""Seeded differential fuzzing of the C ABI against the oracle: random ranks, extents (odd, even,
1), op axes, position pairs, boundary modes and metric broadcast patterns -- the combinations
that steer the host-side dim coalescing, vector/scalar kernel choice, XCD banding, z-banding and
launch splitting.  Bit-exact except contiguous-axis scans/reductions (1e-12)."""

import numpy as np
import pytest

from oracle import refimpl as R

pytestmark = pytest.mark.gpu

PADS = [(1, 0), (0, 1), (1, 1), (0, 0)]
BCS = ["periodic", "fill", "extend"]


@pytest.fixture(scope="module")
def dev():
    from xgcm_amd import device

    return device


def _shape(rng):
    nd = int(rng.integers(1, 6))
    pool = [1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17, 31, 32, 33, 64, 66, 70, 128, 130, 257]
    shape = [int(rng.choice(pool)) for _ in range(nd)]
    while np.prod(shape) > 400_000:
        shape[int(rng.integers(0, nd))] = int(rng.choice([1, 2, 3, 5]))
    return tuple(shape)


def _metric(rng, shape, seed):
    """random broadcast pattern: each dim independently full or 1 (at least the values are positive)."""
    mshape = tuple(s if rng.random() < 0.5 else 1 for s in shape)
    return R.synthetic_metric(mshape, seed)


DTYPES = [np.float64, np.float32]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(12))
def test_fuzz_stencil(dev, seed, dtype):
    rng = np.random.default_rng(1000 + seed)
    for case in range(25):
        shape = _shape(rng)
        axis = int(rng.integers(0, len(shape)))
        op = str(rng.choice(["diff", "interp", "min", "max"]))
        lo, hi = PADS[int(rng.integers(0, 4))]
        if shape[axis] + lo + hi - 1 < 1:
            continue
        bc = str(rng.choice(BCS))
        a = R.synthetic_field(shape, 7000 + 31 * seed + case).astype(dtype)
        if op in ("min", "max") and a.size > 4:
            a.reshape(-1)[rng.integers(0, a.size, 2)] = np.nan
        oshape = list(shape)
        oshape[axis] += lo + hi - 1
        kind = int(rng.integers(0, 4))
        m_in = _metric(rng, shape, 11 + case).astype(dtype) if kind in (2, 3) else None
        m_out = _metric(rng, oshape, 23 + case).astype(dtype) if kind in (1, 3) else None
        fill = float(rng.choice([0.0, 1.5, -2.25]))
        exp = R.stencil1d(op, a, axis, lo, hi, bc, fill, m_in, m_out)
        got = dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, fill, m_in, m_out))
        assert got.shape == exp.shape and got.dtype == exp.dtype == dtype, (shape, axis, op, lo, hi, bc)
        assert np.array_equal(got, exp, equal_nan=True), (shape, axis, op, (lo, hi), bc, kind,
                                                           None if m_in is None else m_in.shape,
                                                           None if m_out is None else m_out.shape)


INT_DTYPES = ["bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64"]


@pytest.mark.parametrize("dtype", INT_DTYPES)
@pytest.mark.parametrize("seed", range(4))
def test_fuzz_integer_lanes(dev, seed, dtype):
    """the integer builds (*_i32: bool / 8 / 16 / 32-bit arrays, *_i64: 64-bit arrays and every scan / sum) over the same
    random shapes: values over the dtype's whole range, so wrap-around, unsigned order and the narrowing are all exercised;
    dtype-exact and bit-exact against numpy on the same integers"""
    rng = np.random.default_rng(5000 + 97 * seed + INT_DTYPES.index(dtype))
    dt = np.dtype(dtype)
    for case in range(12):
        shape = _shape(rng)
        a = rng.integers(0, 2, shape).astype(dt) if dtype == "bool" else \
            rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, shape, dtype=dt, endpoint=True)
        axis = int(rng.integers(0, len(shape)))
        lo, hi = PADS[int(rng.integers(0, 4))]
        bc = str(rng.choice(BCS))
        fill = [0, 1, 3.7, 100][int(rng.integers(0, 4))]
        if shape[axis] + lo + hi - 1 >= 1:
            for op in ("diff", "interp", "min", "max"):
                if dtype == "bool" and op == "diff":
                    continue
                exp = R.stencil1d(op, a, axis, lo, hi, bc, fill)
                got = dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc if (lo or hi) else None, fill))
                assert got.dtype == exp.dtype and np.array_equal(got, exp), (dtype, shape, axis, op, (lo, hi), bc)
        rev = bool(rng.integers(0, 2))
        trims = [(0, 0, 0, 0), (0, 1, 1, 0), (1, 0, 0, 1), (0, 0, 1, 0), (0, 0, 0, 1)][int(rng.integers(0, 5))]
        if shape[axis] - trims[0] - trims[1] >= 1:
            exp = R.cumsum1d(a, axis, *trims, bc, fill, rev, False)
            got = dev.tohost(dev.cumsum1d(a, axis, *trims, bc if (trims[2] or trims[3]) else None, fill, rev, True))
            assert got.dtype == exp.dtype and np.array_equal(got, exp), (dtype, shape, axis, trims, bc, rev)
        got = dev.tohost(dev.reduce1d(a, axis, None, True))
        exp = np.sum(a, axis=axis)
        assert got.dtype == exp.dtype and np.array_equal(got, exp), (dtype, shape, axis)
        w = {axis: (int(rng.integers(0, 3)), int(rng.integers(0, 3)))}
        if shape[axis] >= 2:
            exp = R.pad_nd(a, w, {axis: bc}, {axis: fill})
            got = dev.tohost(dev.pad_nd(a, w, {axis: bc}, {axis: fill}))
            assert got.dtype == exp.dtype and np.array_equal(got, exp), (dtype, shape, w, bc)


@pytest.fixture(params=[1, 2], ids=["default-kernels", "chained-kernels-forced"])
def scan_chain(request):
    """The fuzz shapes are too short for the chained scans / reductions (K5c / K4c) to be picked: run them a second time
    with the chain forced onto every strided march of two chunks or more."""
    from xgcm_amd import _hip

    before = _hip.get_tunable("scan_chain")
    _hip.set_tunable("scan_chain", request.param)
    yield request.param
    _hip.set_tunable("scan_chain", before)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(8))
def test_fuzz_cumsum_reduce(dev, seed, dtype, scan_chain):
    rng = np.random.default_rng(2000 + seed)
    rtol, atol = (1e-12, 1e-9) if dtype == np.float64 else (3e-5, 3e-2)
    tables = [(0, 0, 0, 0), (0, 1, 1, 0), (0, 1, 0, 0), (0, 0, 1, 0), (1, 0, 0, 1), (1, 0, 0, 0), (0, 0, 0, 1)]
    for case in range(20):
        shape = _shape(rng)
        axis = int(rng.integers(0, len(shape)))
        tl, th, pl, ph = tables[int(rng.integers(0, len(tables)))]
        if shape[axis] - tl - th < 1:
            continue
        bc = str(rng.choice(BCS))
        reverse, skipna = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        a = R.synthetic_field(shape, 9000 + 17 * seed + case).astype(dtype)
        if a.size > 4 and rng.random() < 0.5:
            a.reshape(-1)[rng.integers(0, a.size, 2)] = np.nan
        oshape = list(shape)
        oshape[axis] += pl + ph - tl - th
        m_in = _metric(rng, shape, 5 + case).astype(dtype) if rng.random() < 0.4 else None
        m_out = _metric(rng, oshape, 6 + case).astype(dtype) if rng.random() < 0.4 else None
        exp = R.cumsum1d(a, axis, tl, th, pl, ph, bc, 0.5, reverse, skipna, m_in, m_out)
        got = dev.tohost(dev.cumsum1d(a, axis, tl, th, pl, ph, bc, 0.5, reverse, skipna, m_in, m_out))
        contiguous = axis == len(shape) - 1 or all(s == 1 for s in shape[axis + 1:])
        assert got.dtype == dtype
        if contiguous:
            np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol, equal_nan=True)
        else:
            assert np.array_equal(got, exp, equal_nan=True), (shape, axis, (tl, th, pl, ph), bc, reverse, skipna)
        w = _metric(rng, shape, 8 + case).astype(dtype) if rng.random() < 0.5 else None
        exp = R.integrate(a, axis, w, skipna)
        got = dev.tohost(dev.reduce1d(a, axis, w, skipna))
        if contiguous:
            np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol * 10, equal_nan=True)
        else:
            assert np.array_equal(got, exp, equal_nan=True), (shape, axis, "reduce", skipna)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(4))
def test_fuzz_row_scan_any_length(dev, seed, dtype):
    """The vectorised contiguous-axis scan on rows of ANY length: rows that start before a 16-byte boundary
    (lead cells), end after one (tail cells), halo cells inside / next to / outside the aligned groups,
    both directions, every trim / pad combination of the ABI."""
    rng = np.random.default_rng(4000 + seed)
    rtol, atol = (1e-11, 1e-11) if dtype == np.float64 else (3e-4, 3e-4)
    lens = [4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17, 63, 65, 127, 129, 130, 131, 257, 259, 513, 1025, 1030, 2049]
    tables = [(0, 0, 0, 0), (0, 1, 1, 0), (0, 1, 0, 0), (0, 0, 1, 0), (1, 0, 0, 1), (1, 0, 0, 0), (0, 0, 0, 1), (0, 0, 1, 1), (1, 1, 1, 1)]
    for case in range(150):
        nd = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 7)) for _ in range(nd - 1)) + (int(rng.choice(lens)),)
        tl, th, pl, ph = tables[int(rng.integers(0, len(tables)))]
        if shape[-1] - tl - th < 1:
            continue
        a = (rng.random(shape) - 0.5).astype(dtype)
        if rng.random() < 0.3:
            a[rng.random(shape) < 0.1] = np.nan
        reverse, skipna = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        bc = str(rng.choice(["fill", "extend", "periodic"], p=[0.45, 0.45, 0.1])) if (pl or ph) else None
        oshape = list(shape)
        oshape[-1] += pl + ph - tl - th
        m_in = (rng.random(shape) + 0.5).astype(dtype) if rng.random() < 0.3 else None
        m_out = (rng.random(oshape) + 0.5).astype(dtype) if rng.random() < 0.3 else None
        exp = R.cumsum1d(a, nd - 1, tl, th, pl, ph, bc, dtype(0.75), reverse, skipna, m_in, m_out)
        got = dev.tohost(dev.cumsum1d(a, nd - 1, tl, th, pl, ph, bc, 0.75, reverse, skipna, m_in, m_out))
        assert got.dtype == dtype and got.shape == exp.shape
        np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol, equal_nan=True,
                                   err_msg=str((shape, (tl, th, pl, ph), bc, reverse, skipna)))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(4))
def test_fuzz_two_axis_and_vorticity(dev, seed, dtype):
    rng = np.random.default_rng(3000 + seed)
    for case in range(15):
        nd = int(rng.integers(2, 5))
        shape = tuple(int(rng.choice([1, 2, 3, 5])) for _ in range(nd - 2)) + (int(rng.choice([1, 2, 3, 4, 5, 9, 33])),
                                                                              int(rng.choice([2, 4, 6, 64, 130, 258])))
        a = R.synthetic_field(shape, 100 + case).astype(dtype)
        b = R.synthetic_field(shape, 200 + case).astype(dtype)
        op = str(rng.choice(["diff", "interp", "min", "max"]))
        order = int(rng.integers(0, 2))
        padx, pady = [(1, 0), (0, 1)][int(rng.integers(0, 2))], [(1, 0), (0, 1)][int(rng.integers(0, 2))]
        bcx, bcy = str(rng.choice(BCS)), str(rng.choice(BCS))
        ax_x, ax_y = nd - 1, nd - 2
        if order == 0:
            exp = R.stencil1d(op, R.stencil1d(op, a, ax_x, *padx, bcx, 0.5), ax_y, *pady, bcy, -1.0)
        else:
            exp = R.stencil1d(op, R.stencil1d(op, a, ax_y, *pady, bcy, -1.0), ax_x, *padx, bcx, 0.5)
        if dev.stencil2d_supported(a, padx, pady):
            got = dev.tohost(dev.stencil2d(op, a, order, padx, bcx, 0.5, pady, bcy, -1.0))
            assert got.dtype == dtype and np.array_equal(got, exp, equal_nan=True), (shape, op, order, padx, pady, bcx, bcy)
        area = R.synthetic_metric((1,) * (nd - 2) + shape[-2:], 300 + case).astype(dtype) if rng.random() < 0.7 else None
        exp = R.vorticity(a, b, area if area is not None else np.ones((1,) * nd, dtype=dtype), bcx, bcy, dtype(0.25), dtype(-0.5))
        got = dev.tohost(dev.vorticity(a, b, area, bcx, bcy, 0.25, -0.5))
        assert np.array_equal(got, exp), (shape, "vorticity", bcx, bcy, area is not None)
        exp = R.divergence(a, b, area if area is not None else np.ones((1,) * nd, dtype=dtype), bcx, bcy, dtype(0.25), dtype(-0.5))
        got = dev.tohost(dev.divergence(a, b, area, bcx, bcy, 0.25, -0.5))
        assert np.array_equal(got, exp), (shape, "divergence", bcx, bcy, area is not None)
        ex, ey = R.gradient(a, bcx, bcy, dtype(0.25), dtype(-0.5), area, None)
        gx, gy = dev.gradient(a, bcx, bcy, 0.25, -0.5, area, None)
        assert np.array_equal(dev.tohost(gx), ex, equal_nan=True) and np.array_equal(dev.tohost(gy), ey, equal_nan=True), (shape, "gradient", bcx, bcy)
        ex, ey = R.flux(a, b, b, bcx, bcy, dtype(0.25), dtype(-0.5))
        fx, fy = dev.flux(a, b, b, bcx, bcy, 0.25, -0.5)
        assert np.array_equal(dev.tohost(fx), ex, equal_nan=True) and np.array_equal(dev.tohost(fy), ey, equal_nan=True), (shape, "flux", bcx, bcy)
