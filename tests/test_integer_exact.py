"""Integer and bool arrays: the results numpy (hence the reference) returns, bit for bit and dtype for dtype.

The reference's bodies are plain numpy expressions run in the array's own dtype (xgcm/gridops.py:23-24,76-77,123-126,
172-175,227-278) after `numpy.pad`, which keeps the dtype and casts the fill value (xgcm/padding.py:610-615): integers
wrap modulo 2^bits, `interp` leaves through `/ 2.0`, `cumsum` accumulates in int64 / uint64, bool `diff` raises.  The
backend serves them on int64 lanes between xg_convert calls (xgcm_amd.dtypes, device._int_*).

Fixtures: tests/golden/gridops_vectors_int.npz holds the outputs of the reference's OWN 40 ufunc bodies on seeded
bool / int8..int64 / uint8..uint64 inputs (dtype extremes, 2^53 + 1, 2^62, uint64 above 2^63), written by
oracle/make_golden.py.  Every test runs on three backends: the numpy oracle double, the host build of the C ABI (CPU,
the real `*_i64` + `xg_convert` symbols through the product's dtype policy) and -- marked gpu -- the HIP library.
"""

import os

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid, gridops
from xgcm_amd import dtypes as DT

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
VEC = np.load(os.path.join(GOLDEN, "gridops_vectors_int.npz"))
FVEC = np.load(os.path.join(GOLDEN, "gridops_vectors.npz"))
INT_DTYPES = ("bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64")
INT_FILL = 3.7   # as in oracle/make_golden.py
MODES = ("periodic", "fill", "extend")
UFUNCS = sorted(n for n, f in vars(gridops).items() if isinstance(f, gridops.HipGridUFunc))


@pytest.fixture(params=["oracle-double", "host-abi", pytest.param("hip", marks=pytest.mark.gpu)])
def ibackend(request, monkeypatch):
    if request.param == "oracle-double":
        from oracle import fake_device

        fake_device.install(monkeypatch)
    elif request.param == "host-abi":
        import host_abi_device

        host_abi_device.install(monkeypatch)
    return request.param


def _same(got, want):
    """bit-exact AND dtype-exact"""
    got = np.asarray(got)
    assert got.dtype == want.dtype, f"dtype {got.dtype}, reference {want.dtype}"
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def _split(name):
    func, rest = name.split("_", 1)
    f, t = rest.split("_to_")
    return func, f, t


def _grid(n, from_pos, to_pos, padding):
    """one axis X with exactly the two positions of a ufunc, after (outer dims) Z, Y"""
    lengths = {"center": n, "left": n, "right": n, "inner": n - 1, "outer": n + 1}
    length_of_center = {v: k for k, v in lengths.items()}
    del length_of_center
    coords = {"Z": ("Z", np.arange(3.0)), "Y": ("Y", np.arange(4.0))}
    for pos in (from_pos, to_pos):
        coords[f"x_{pos}"] = (f"x_{pos}", np.arange(float(lengths[pos])))
    return Grid(Dataset(coords=coords), coords={"X": {from_pos: f"x_{from_pos}", to_pos: f"x_{to_pos}"}}, padding=padding,
                autoparse_metadata=False)


# ----------------------------------------------------------------------------------------------
# the dtype policy on its own (host logic)
# ----------------------------------------------------------------------------------------------
def test_policy_matches_numpy_promotion():
    for name in INT_DTYPES:
        dt = np.dtype(name)
        a = np.ones(4, dtype=dt)
        assert DT.cumsum_dtype(dt) == np.cumsum(a).dtype == np.sum(a).dtype
        for op in ("diff", "interp", "min", "max"):
            if name == "bool" and op == "diff":
                with pytest.raises(TypeError, match="numpy boolean subtract"):
                    DT.stencil_plan(op, dt)
                continue
            plan = DT.stencil_plan(op, dt)
            assert plan.lanes == "int" and plan.result == R.stencil1d(op, a, 0, 0, 0, None).dtype
            assert plan.compute == (np.int64 if dt.itemsize == 8 else np.int32)  # 64-bit types on int64 lanes, the rest on int32
            assert plan.unsigned == (name in ("uint64", "uint32") and op in ("min", "max"))
        for mdt in (np.float32, np.float64):
            m = np.ones(4, dtype=mdt)
            assert DT.stencil_plan("diff", dt, mdt, mdt).compute == (a * m).dtype  # metric first: float lanes
            assert DT.stencil_plan("diff", dt, None, mdt).divide_as == (a / m).dtype if name != "bool" else True
        for other in INT_DTYPES:
            for op, f in (("mul", np.multiply), ("add", np.add), ("div", np.divide)):
                lanes, rt = DT.binary_plan(op, dt, np.dtype(other))
                assert rt == f(a, np.ones(4, dtype=other)).dtype, (name, other, op)
    assert DT.float_of(np.float32, np.float32) == np.float32 and DT.float_of(np.int8, np.float32) == np.float32
    assert DT.float_of(np.int32, np.float32) == np.float64 and DT.float_of(np.float32, np.float64) == np.float64


def test_fill_value_is_cast_like_numpy_pad():
    assert DT.fill_as(np.int8, 3.7) == 3 and DT.fill_as(np.int8, -1.5) == -1 and DT.fill_as(np.uint8, -1.5) == 255
    assert DT.fill_as(np.bool_, 2.7) == True and DT.fill_as(np.int64, 2**60 + 1) == 2**60 + 1  # noqa: E712
    assert DT.fill_as(np.float64, 3) == 3.0 and DT.fill_as(np.int32, None) == 0
    with pytest.raises(OverflowError):
        DT.fill_as(np.int8, 300)
    with pytest.raises(ValueError):
        DT.fill_as(np.int16, np.nan)


# ----------------------------------------------------------------------------------------------
# the oracle against the reference's outputs on integers (pins numpy == reference here)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", UFUNCS)
def test_oracle_equals_reference_bodies_on_integers(name):
    func, f, t = _split(name)
    for dtype in INT_DTYPES:
        a = VEC[f"{name}|{dtype}|in"]
        assert a.dtype == np.dtype(dtype)
        for bc in MODES:
            want = VEC[f"{name}|{dtype}|{bc}"]
            if want.ndim == 0:  # the reference body raised
                with pytest.raises(TypeError):
                    R.stencil1d(func, a, 2, *R.STENCIL_PADDING_WIDTH[(f, t)], bc, INT_FILL)
                continue
            if func == "cumsum":
                (lo, hi), _, _, drop_last = R.CUMSUM_UFUNC_TABLE[(f, t)]
                got = R.cumsum1d(a, 2, 0, 1 if drop_last else 0, lo, hi, bc, INT_FILL, reverse=False, skipna=False)
            else:
                got = R.stencil1d(func, a, 2, *R.STENCIL_PADDING_WIDTH[(f, t)], bc, INT_FILL)
            _same(got, want)


# ----------------------------------------------------------------------------------------------
# the reference's vectors straight through the 40 ufunc bodies and the fused calls
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", UFUNCS)
def test_reference_integer_vectors_through_the_ufunc_bodies(ibackend, name):
    """`HipGridUFunc.ufunc` (the raw plugin level: already padded array, core dim last) and the fused `__call__`
    (halo inside the kernel) on the reference-generated integer fixtures: bit-exact, dtype-exact."""
    uf = getattr(gridops, name)
    func, f, t = _split(name)
    lo, hi = uf.padding_width["X"]
    for dtype in INT_DTYPES:
        a = VEC[f"{name}|{dtype}|in"]
        for bc, mode in (("periodic", "wrap"), ("fill", "constant"), ("extend", "edge")):
            want = VEC[f"{name}|{dtype}|{bc}"]
            kw = {"constant_values": INT_FILL} if mode == "constant" else {}
            grid = _grid(a.shape[-1] - {"center": 0, "left": 0, "right": 0, "inner": -1, "outer": 1}[f], f, t, bc)
            da = DataArray(a, ("Z", "Y", f"x_{f}"))
            if want.ndim == 0:
                with pytest.raises(TypeError, match="numpy boolean subtract"):
                    uf.ufunc(np.pad(a, [(0, 0), (0, 0), (lo, hi)], mode, **kw))
                with pytest.raises(TypeError, match="numpy boolean subtract"):
                    uf(grid, da, axis=[("X",)], padding=bc, fill_value=INT_FILL)
                continue
            if uf.pad_before_func:
                raw = uf.ufunc(np.pad(a, [(0, 0), (0, 0), (lo, hi)], mode, **kw))
            else:  # cumsum family: the body, then the pad of the cumulative values
                raw = np.pad(np.asarray(uf.ufunc(a)), [(0, 0), (0, 0), (lo, hi)], mode, **kw)
            _same(raw, want)
            fused = uf(grid, da, axis=[("X",)], padding=bc, fill_value=INT_FILL)
            assert fused.dims == ("Z", "Y", f"x_{t}")
            _same(fused.values, want)
            # the same operator along a non-last (strided) axis of the transposed field
            dat = DataArray(np.ascontiguousarray(np.moveaxis(a, 2, 0)), (f"x_{f}", "Z", "Y"))
            fused_t = uf(grid, dat, axis=[("X",)], padding=bc, fill_value=INT_FILL)
            _same(np.moveaxis(fused_t.values, 0, 2), want)


@pytest.mark.parametrize("name", UFUNCS)
def test_reference_float_vectors_through_the_ufunc_bodies(ibackend, name):
    """the float64 fixtures of the reference bodies, likewise straight through `.ufunc` and the fused call"""
    uf = getattr(gridops, name)
    func, f, t = _split(name)
    lo, hi = uf.padding_width["X"]
    a = FVEC[f"{name}|in"]
    for bc, mode in (("periodic", "wrap"), ("fill", "constant"), ("extend", "edge")):
        want = FVEC[f"{name}|{bc}"]
        kw = {"constant_values": 1.25} if mode == "constant" else {}
        if uf.pad_before_func:
            raw = uf.ufunc(np.pad(a, [(0, 0), (0, 0), (lo, hi)], mode, **kw))
        else:
            raw = np.pad(np.asarray(uf.ufunc(a)), [(0, 0), (0, 0), (lo, hi)], mode, **kw)
        exact = func != "cumsum" or ibackend == "oracle-double"  # contiguous-axis scans are re-associated by contract
        if exact:
            _same(raw, want)
        else:
            np.testing.assert_allclose(raw, want, rtol=1e-12, atol=1e-13)
        grid = _grid(a.shape[-1] - {"center": 0, "left": 0, "right": 0, "inner": -1, "outer": 1}[f], f, t, bc)
        fused = uf(grid, DataArray(a, ("Z", "Y", f"x_{f}")), axis=[("X",)], padding=bc, fill_value=1.25)
        if exact:
            _same(fused.values, want)
        else:
            np.testing.assert_allclose(fused.values, want, rtol=1e-12, atol=1e-13)


# ----------------------------------------------------------------------------------------------
# the public Grid API: the cases that used to come back float64-rounded
# ----------------------------------------------------------------------------------------------
def _xgrid(n, padding="periodic", fill_value=None):
    ds = Dataset(coords={"xc": ("xc", np.arange(n) + 0.5), "xg": ("xg", np.arange(n) * 1.0)})
    return Grid(ds, coords={"X": {"center": "xc", "left": "xg"}}, padding=padding, fill_value=fill_value,
                autoparse_metadata=False)


def test_int64_beyond_2_53_stays_exact(ibackend):
    a = np.arange(2**53 + 1, 2**53 + 9, dtype=np.int64)
    grid = _xgrid(8)
    d = grid.diff(DataArray(a, ("xc",)), "X")
    _same(d.values, np.array([-7, 1, 1, 1, 1, 1, 1, 1], dtype=np.int64))
    c = grid.cumsum(DataArray(np.full(8, 2**53 + 1, dtype=np.int64), ("xc",)), "X", to="left", padding="fill")
    _same(c.values, R.grid_cumsum(np.full(8, 2**53 + 1, dtype=np.int64), 0, "center", "left", "fill", skipna=False))
    assert int(grid.cumsum(DataArray(np.full(8, 2**53 + 1, dtype=np.int64), ("xg",)), "X").values[-1]) == 72057594037927944
    _same(grid.max(DataArray(a, ("xc",)), "X").values, R.stencil1d("max", a, 0, 1, 0, "periodic"))
    _same(grid.interp(DataArray(a, ("xc",)), "X").values, R.stencil1d("interp", a, 0, 1, 0, "periodic"))


def test_unsigned_wraps_and_keeps_its_dtype(ibackend):
    a = np.array([5, 2, 0, 9, 200, 250, 3, 7], dtype=np.uint8)
    grid = _xgrid(8)
    _same(grid.diff(DataArray(a, ("xc",)), "X").values, R.stencil1d("diff", a, 0, 1, 0, "periodic"))
    assert grid.diff(DataArray(a, ("xc",)), "X").values.dtype == np.uint8
    _same(grid.interp(DataArray(a, ("xc",)), "X").values, R.stencil1d("interp", a, 0, 1, 0, "periodic"))  # sum wraps at 8 bits
    _same(grid.cumsum(DataArray(a, ("xc",)), "X", padding="extend").values,
          R.grid_cumsum(a, 0, "center", "left", "extend", skipna=False))
    big = np.array([2**64 - 1, 2**63 + 5, 3, 2**63 - 1, 0, 2**64 - 2, 7, 2**63], dtype=np.uint64)
    for op in ("diff", "min", "max", "interp"):
        _same(getattr(grid, op)(DataArray(big, ("xc",)), "X", padding="fill", fill_value=2**63 + 1).values,
              R.stencil1d(op, big, 0, 1, 0, "fill", 2**63 + 1))


def test_32_bit_arrays_run_on_their_own_lanes(ibackend):
    """int32 / uint32 compute on int32 lanes as they are (no widening): wrap at 32 bits, unsigned order for min / max,
    numpy.pad's cast of the fill value; mixed-width labelled arithmetic promotes like numpy before it reaches the lanes"""
    grid = _xgrid(8)
    u = np.array([2**32 - 1, 2**31 + 5, 3, 2**31 - 1, 0, 2**32 - 2, 7, 2**31], dtype=np.uint32)
    i = np.array([2**31 - 1, -2**31, 3, -1, 0, 2**31 - 2, -7, 12], dtype=np.int32)
    for a, fill in ((u, 2**31 + 1), (i, -5)):
        for op in ("diff", "min", "max", "interp"):
            for pad in ("fill", "periodic", "extend"):
                got = getattr(grid, op)(DataArray(a, ("xc",)), "X", padding=pad, fill_value=fill).values
                _same(got, R.stencil1d(op, a, 0, 1, 0, pad, fill))
        _same(grid.cumsum(DataArray(a, ("xc",)), "X", padding="fill", fill_value=0).values,
              R.grid_cumsum(a, 0, "center", "left", "fill", 0, skipna=False))
    da = DataArray(i.reshape(2, 4), ("y", "x"))
    for other in (np.int8, np.uint16, np.uint32, np.int64, np.uint8):
        b = (np.arange(8).reshape(2, 4) * 37 - 100).astype(other)
        db = DataArray(b, ("y", "x"))
        for f, g in ((lambda p, q: p + q, np.add), (lambda p, q: p - q, np.subtract), (lambda p, q: p * q, np.multiply)):
            _same(f(da, db).values, g(i.reshape(2, 4), b))
            _same(f(db, da).values, g(b, i.reshape(2, 4)))
    n8, u16 = DataArray(np.array([-128, -1, 0, 127], dtype=np.int8), ("x",)), DataArray(np.array([65535, 1, 0, 40000], dtype=np.uint16), ("x",))
    _same((n8 * u16).values, n8.values * u16.values)  # int8 next to uint16 -> int32: sign- and zero-extended into the lanes
    _same((n8 - n8).values, n8.values - n8.values)


def test_bool_follows_numpy(ibackend):
    b = np.array([True, False, False, True, True, False, True, False])
    grid = _xgrid(8)
    with pytest.raises(TypeError, match="numpy boolean subtract"):
        grid.diff(DataArray(b, ("xc",)), "X")
    for op in ("min", "max", "interp"):
        _same(getattr(grid, op)(DataArray(b, ("xc",)), "X").values, R.stencil1d(op, b, 0, 1, 0, "periodic"))
    _same(grid.cumsum(DataArray(b, ("xc",)), "X", padding="fill", fill_value=1).values,
          R.grid_cumsum(b, 0, "center", "left", "fill", 1, skipna=False))


def test_fill_value_into_an_integer_field(ibackend):
    a = np.arange(-4, 4, dtype=np.int16)
    grid = _xgrid(8, padding="fill", fill_value=2.9)
    _same(grid.diff(DataArray(a, ("xc",)), "X").values, R.stencil1d("diff", a, 0, 1, 0, "fill", 2.9))  # 2.9 -> 2
    with pytest.raises(OverflowError):
        grid.diff(DataArray(a.astype(np.int8), ("xc",)), "X", fill_value=300)


def test_metrics_promote_like_numpy(ibackend):
    rng = np.random.default_rng(5)
    nz, ny, nx = 3, 6, 8
    T = rng.integers(-2**40, 2**40, (nz, ny, nx), dtype=np.int64)
    T[0, 0, :3] = [2**53 + 1, 2**53 + 3, -(2**53) - 5]
    dx = rng.random((ny, nx)) + 1.0
    dz = rng.random(nz) + 1.0
    ds = Dataset({"T": (("Z", "YC", "XC"), T), "dxC": (("YC", "XG"), dx), "dxT": (("YC", "XC"), dx + 0.5), "drF": (("Z",), dz)},
                 coords={"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) * 1.0),
                         "Z": ("Z", np.arange(nz) * 1.0), "Zl": ("Zl", np.arange(nz) - 0.5)})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Z": "fill"}, metrics={("X",): ["dxC", "dxT"], ("Z",): ["drF"]}, autoparse_metadata=False)
    # derivative: the difference is taken in int64 (exact), then promoted and divided
    _same(grid.derivative(ds["T"], "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx[None]))
    # metric_weighted: `T * dx` is float64 before the operator
    _same(grid.interp(ds["T"], "X", metric_weighted="X").values,
          R.stencil1d("interp", T, 2, 1, 0, "periodic", m_in=(dx + 0.5)[None], m_out=dx[None]))
    _same(grid.integrate(ds["T"], "Z").values, R.integrate(T, 0, dz[:, None, None]))
    _same(grid.cumint(ds["T"], "Z", padding="fill").values,
          R.grid_cumsum(T * dz[:, None, None], 0, "center", "left", "fill"))
    small = T.astype(np.int16)
    _same(grid.derivative(DataArray(small, ("Z", "YC", "XC")), "X").values,
          R.stencil1d("diff", small, 2, 1, 0, "periodic", m_out=dx[None]))


def test_two_axes_and_labelled_arithmetic_keep_integers_integral(ibackend):
    rng = np.random.default_rng(6)
    a = rng.integers(-100, 100, (3, 6, 8), dtype=np.int32)
    ds = Dataset(coords={"XC": ("XC", np.arange(8) + 0.5), "XG": ("XG", np.arange(8) * 1.0), "YC": ("YC", np.arange(6) + 0.5),
                         "YG": ("YG", np.arange(6) * 1.0), "Z": ("Z", np.arange(3.0))})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                padding={"X": "periodic", "Y": "extend"}, autoparse_metadata=False)
    da = DataArray(a, ("Z", "YC", "XC"))
    want = R.stencil1d("diff", R.stencil1d("diff", a, 2, 1, 0, "periodic"), 1, 1, 0, "extend")
    _same(grid.diff(da, ["X", "Y"]).values, want)
    wi = R.stencil1d("interp", R.stencil1d("interp", a, 2, 1, 0, "periodic"), 1, 1, 0, "extend")
    _same(grid.interp(da, ["X", "Y"]).values, wi)
    _same((da + da).values, a + a)
    _same((da * 3).values, a * 3)
    _same((da * 0.5).values, a * 0.5)
    _same((da - DataArray(a[0].astype(np.int8), ("YC", "XC"))).values, a - a[0].astype(np.int8))
    _same((da / da).values, a / a) if not (a == 0).any() else None
    _same(da.sum("Z").values, a.sum(0))
    _same(da.cumsum("XC").values, np.cumsum(a, 2))
    gx, gy = grid.gradient(da)
    _same(gx.values, R.stencil1d("diff", a, 2, 1, 0, "periodic"))
    _same(gy.values, R.stencil1d("diff", a, 1, 1, 0, "extend"))


# ----------------------------------------------------------------------------------------------
# GPU only: the vector kernels (aligned rows, long columns) and device-resident integer tensors
# ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", INT_DTYPES)
def test_integer_kernels_at_kernel_sizes(dtype):
    """shapes that reach the 16-byte-lane kernels (K1, K2S / K2Sy, the marching and chained scans, the aligned-group
    contiguous scan) and their scalar-lane twins (odd extents), against numpy on the same integers"""
    import torch

    from xgcm_amd import device as dev

    rng = np.random.default_rng(hash(dtype) % 2**31)
    dt = np.dtype(dtype)
    for shape in ((5, 260, 512), (3, 67, 131), (2, 300, 64)):
        a = rng.integers(0, 2, shape).astype(dt) if dtype == "bool" else \
            rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, shape, dtype=dt, endpoint=True)
        t = torch.from_numpy(a).cuda()
        for ax in (0, 1, 2):
            for (lo, hi) in ((1, 0), (0, 1), (1, 1), (0, 0)):
                for bc in ("periodic", "fill", "extend"):
                    for op in ("diff", "interp", "min", "max"):
                        if dtype == "bool" and op == "diff":
                            continue
                        got = dev.stencil1d(op, t, ax, lo, hi, bc if (lo or hi) else None, 3.7)
                        assert got.is_cuda
                        _same(dev.tohost(got), R.stencil1d(op, a, ax, lo, hi, bc, 3.7))
            for rev in (False, True):
                for trims in ((0, 0, 0, 0), (0, 1, 1, 0), (1, 0, 0, 1), (0, 0, 1, 0)):
                    for bc in ("periodic", "fill", "extend"):
                        got = dev.cumsum1d(t, ax, *trims, bc if (trims[2] or trims[3]) else None, 3.7, rev, True)
                        _same(dev.tohost(got), R.cumsum1d(a, ax, *trims, bc, 3.7, rev, False))
            _same(dev.tohost(dev.reduce1d(t, ax, None, True)), np.sum(a, axis=ax))
        _same(dev.tohost(dev.pad_nd(t, {0: (2, 1), 2: (1, 3)}, {0: "periodic", 2: "fill"}, {2: 5.5})),
              R.pad_nd(a, {0: (2, 1), 2: (1, 3)}, {0: "periodic", 2: "fill"}, {2: 5.5}))


@pytest.mark.gpu
def test_convert_kernel_is_numpy_astype():
    import torch

    from xgcm_amd import device as dev

    rng = np.random.default_rng(9)
    n = 4099  # odd: vector body + scalar tail
    for src in INT_DTYPES + ("float32", "float64"):
        sdt = np.dtype(src)
        if sdt.kind == "f":
            a = (rng.random(n) * 200 - 100).astype(sdt)
        elif src == "bool":
            a = rng.integers(0, 2, n).astype(sdt)
        else:
            a = rng.integers(np.iinfo(sdt).min, np.iinfo(sdt).max, n, dtype=sdt, endpoint=True)
        for dst in INT_DTYPES + ("float32", "float64"):
            if sdt.kind == "f" and np.dtype(dst).kind in "iu" and np.dtype(dst).itemsize < 8:
                continue  # out-of-range float -> narrow int is undefined in C and in numpy alike
            with np.errstate(invalid="ignore"):
                want = a.astype(dst)
            if sdt.kind == "f" and np.dtype(dst).kind == "u":
                continue  # negative float -> unsigned: undefined
            got = dev.tohost(dev.convert(torch.from_numpy(a).cuda(), dst))
            _same(got, want)
    w = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    _same(dev.tohost(dev.convert(torch.from_numpy(w).cuda(), np.float64, via=np.int8, scale=0.5)), w.astype(np.int8) / 2.0)
    u = rng.integers(0, 2**64 - 1, n, dtype=np.uint64, endpoint=True)
    flipped = dev.convert(torch.from_numpy(u).cuda(), np.int64, flip=True)
    assert np.array_equal(np.argsort(dev.tohost(flipped), kind="stable"), np.argsort(u, kind="stable"))
    _same(dev.tohost(dev.convert(flipped, np.uint64, flip=True)), u)
