"""float32 fields: the reference computes in the input's dtype (numpy), so f32 in -> f32 out with
numpy-float32 arithmetic.  Kernel-level parity of the *_f32 ABI entry points against the oracle
run on float32 arrays (bit-exact except contiguous-axis scans/reductions), and the Grid surface
on both backends.  Mixed float32/float64 operands promote to float64 like numpy/xarray."""

import itertools

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid

PADS = [(1, 0), (0, 1), (1, 1), (0, 0)]
BCS = ["periodic", "fill", "extend"]
F = np.float32


def f32(shape, seed, nan=False):
    a = R.synthetic_field(shape, seed).astype(F)
    if nan and a.size > 3:
        a.reshape(-1)[[1, a.size // 2, a.size - 1]] = np.nan
    return a


def m32(shape, keep, seed):
    return R.synthetic_metric([s if d in keep else 1 for d, s in enumerate(shape)], seed).astype(F)


def _eq(a, b):
    assert a.dtype == b.dtype == F, (a.dtype, b.dtype)
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.fixture(scope="module")
def dev():
    from xgcm_amd import device

    return device


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(6, 10, 128), (3, 7, 33), (2, 5, 4, 66), (257,), (5, 1, 66)])
@pytest.mark.parametrize("op", ["diff", "interp", "min", "max"])
def test_f32_stencil_all_axes_pads_bcs(dev, shape, op):
    a = f32(shape, 11, nan=op in ("min", "max"))
    for axis in range(len(shape)):
        for (lo, hi), bc in itertools.product(PADS, BCS):
            if shape[axis] + lo + hi - 1 < 1:
                continue
            _eq(dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, 1.25)), R.stencil1d(op, a, axis, lo, hi, bc, 1.25))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(5, 6, 64), (3, 37, 130), (2, 3, 4, 34)])
def test_f32_metric_weighted_bitwise(dev, shape):
    nd = len(shape)
    a = f32(shape, 5)
    for op, axis in itertools.product(["diff", "interp"], range(nd)):
        for (lo, hi), bc in itertools.product(PADS, BCS):
            n_out = shape[axis] + lo + hi - 1
            if n_out < 1:
                continue
            oshape = list(shape)
            oshape[axis] = n_out
            for keep in (set(range(nd)), {axis}, {nd - 1, max(nd - 2, 0)}):
                mi, mo = m32(shape, keep, 31), m32(oshape, keep, 32)
                _eq(dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, 0.75, mi, mo)),
                    R.stencil1d(op, a, axis, lo, hi, bc, 0.75, mi, mo))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(6, 10, 128), (3, 7, 33), (300,), (3, 700)])
def test_f32_cumsum_and_reduce(dev, shape):
    a = f32(shape, 7, nan=True)
    nd = len(shape)
    for axis in range(nd):
        last = axis == nd - 1
        for reverse, skipna in itertools.product([False, True], [True, False]):
            for tl, th, pl, ph in [(0, 0, 0, 0), (0, 1, 1, 0), (0, 0, 1, 0), (1, 0, 0, 1)]:
                if shape[axis] - tl - th < 1:
                    continue
                for bc in BCS:
                    exp = R.cumsum1d(a, axis, tl, th, pl, ph, bc, 0.5, reverse, skipna)
                    got = dev.tohost(dev.cumsum1d(a, axis, tl, th, pl, ph, bc, 0.5, reverse, skipna))
                    assert got.dtype == F
                    if last and shape[axis] > 1:
                        np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True)
                    else:
                        _eq(got, exp)
        for skipna in (True, False):
            w = m32(shape, {axis}, 21)
            exp = R.integrate(a, axis, w, skipna)
            got = dev.tohost(dev.reduce1d(a, axis, w, skipna))
            assert got.dtype == F
            if last:
                np.testing.assert_allclose(got, exp, rtol=2e-5, atol=1e-2, equal_nan=True)
            else:
                _eq(got, exp)


@pytest.mark.gpu
def test_f32_pad_binary_vorticity_stencil2d_synthetic(dev):
    a = f32((4, 5, 6), 17)
    for widths, bc, fill in [({2: (1, 1)}, {2: "periodic"}, {}), ({2: (0, 1), 1: (2, 0)}, {2: "periodic", 1: "fill"}, {1: 1.5}),
                             ({0: (7, 9), 2: (13, 3)}, {0: "periodic", 2: "extend"}, {})]:
        _eq(dev.tohost(dev.pad_nd(a, widths, bc, fill)), R.pad_nd(a, widths, bc, fill))
    a = f32((3, 4, 6, 10), 19)
    for op in ("mul", "div", "add", "sub"):
        for bshape in [(3, 4, 6, 10), (1, 1, 6, 10), (1, 4, 1, 1), (1, 1, 1, 1)]:
            b = R.synthetic_metric(bshape, 23).astype(F)
            _eq(dev.tohost(dev.binary(op, a, b)), R.binary(op, a, b))
    for shape in [(3, 9, 64), (2, 70, 34)]:
        u, v = f32(shape, 51), f32(shape, 52)
        area = R.synthetic_metric((1,) + shape[-2:], 53).astype(F)
        for bx, by in itertools.product(BCS, BCS):
            _eq(dev.tohost(dev.vorticity(u, v, area, bx, by, 0.25, -0.5)), R.vorticity(u, v, area, bx, by, F(0.25), F(-0.5)))
            for order in (0, 1):
                if not dev.stencil2d_supported(u, (1, 0), (1, 0)):
                    assert shape[-1] % 4  # float lanes hold 4 elements: the Grid layer then runs the axes one by one
                    continue
                t = R.stencil1d("interp", u, 2 if order == 0 else 1, 1, 0, bx if order == 0 else by, 0.5)
                exp = R.stencil1d("interp", t, 1 if order == 0 else 2, 1, 0, by if order == 0 else bx, 0.5)
                _eq(dev.tohost(dev.stencil2d("interp", u, order, (1, 0), bx, 0.5, (1, 0), by, 0.5)), exp)
    import torch

    got = dev.tohost(dev.synthetic((1000,), 4, 77, dtype=torch.float32))
    _eq(got, R.synthetic(1000, 4, 77).astype(F))


@pytest.mark.gpu
def test_mixed_dtypes_promote_to_float64(dev):
    a32, m64 = f32((4, 6, 32), 3), R.synthetic_metric((1, 6, 32), 31)
    got = dev.tohost(dev.stencil1d("diff", a32, 2, 1, 0, "periodic", 0.0, None, m64))
    exp = R.stencil1d("diff", a32, 2, 1, 0, "periodic", 0.0, None, m64)  # numpy: the float32 difference, then / float64
    assert got.dtype == np.float64 and np.array_equal(got, exp)
    got = dev.tohost(dev.stencil1d("diff", a32, 2, 1, 0, "periodic", 0.0, m64, m64))  # float64 product first: all float64
    assert got.dtype == np.float64 and np.array_equal(got, R.stencil1d("diff", a32, 2, 1, 0, "periodic", 0.0, m64, m64))
    assert dev.tohost(dev.binary("mul", a32, m64)).dtype == np.float64
    ints = np.arange(24).reshape(4, 6)
    out = dev.tohost(dev.stencil1d("diff", ints, 1, 1, 0, "fill"))
    assert out.dtype == ints.dtype and np.array_equal(out, R.stencil1d("diff", ints, 1, 1, 0, "fill"))  # integers stay integral
    got = dev.tohost(dev.stencil1d("diff", ints, 1, 1, 0, "fill", 0.0, m64[0, :4, :6], None))  # int * float64 metric -> float64
    assert got.dtype == np.float64 and np.array_equal(got, R.stencil1d("diff", ints, 1, 1, 0, "fill", 0.0, m64[0, :4, :6]))


def test_f32_grid_surface(backend):
    """Grid ops on float32 fields return float32 computed in float32, like the reference's numpy path."""
    nz, ny, nx = 4, 10, 16
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) + 0.5), "Zl": ("Zl", np.arange(nz) * 1.0)}
    T = f32((nz, ny, nx), 2)
    dx, drF = R.synthetic_metric((ny, nx), 31).astype(F), R.synthetic_metric((nz,), 33).astype(F)
    ds = Dataset({"T": (("Z", "YC", "XC"), T), "dxC": (("YC", "XG"), dx), "drF": (("Z",), drF)}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, metrics={("X",): ["dxC"], ("Z",): ["drF"]},
                autoparse_metadata=False)
    d = grid.diff(ds["T"], "X")
    assert d.values.dtype == F and np.array_equal(d.values, R.stencil1d("diff", T, 2, 1, 0, "periodic"))
    i2 = grid.interp(ds["T"], ["X", "Y"])
    want = R.stencil1d("interp", R.stencil1d("interp", T, 2, 1, 0, "periodic"), 1, 1, 0, "extend")
    assert i2.values.dtype == F and np.array_equal(i2.values, want)
    dv = grid.derivative(ds["T"], "X")
    assert dv.values.dtype == F and np.array_equal(dv.values, R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx[None]))
    c = grid.cumsum(ds["T"], "Z")
    assert c.values.dtype == F and np.array_equal(c.values, R.grid_cumsum(T, 0, "center", "left", "fill"))
    s = grid.integrate(ds["T"], "Z")
    assert s.values.dtype == F and np.array_equal(s.values, R.integrate(T, 0, drF[:, None, None]))
    half = ds["T"] * 0.5  # weak python scalar keeps float32
    assert half.values.dtype == F and np.array_equal(half.values, T * F(0.5))
