"""Arrays as files and model output hold them: non-native byte order, float16, and dtypes nobody serves.

The reference's bodies are numpy expressions that take any numpy dtype (xgcm/gridops.py:23-24,76-77; `np.pad` keeps the
dtype, xgcm/padding.py:610-615).  MITgcm's MDS files are big-endian; `np.fromfile(f, ">f4")` and xmitgcm arrays go into
`Grid` as they are, and numpy returns NATIVE results holding the same values.  Here such an array crosses PCIe as raw
bytes and is byte-swapped in HBM (`xg_bswap`); the one intake rule is `xgcm_amd.dtypes.host_intake`, shared by the
product (`device._raw_device`) and the host-ABI double.  Expected values below are numpy's own, computed ON the
non-native arrays (numpy reads the byte order from the dtype), never on a pre-swapped copy.

Every test runs on three backends: the numpy oracle double, the host build of the C ABI, and -- marked gpu -- HIP.
"""

import numpy as np
import pytest

from oracle import refimpl as R
from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd import dtypes as DT

NON_NATIVE = (">f4", ">f8", ">i4", ">i2", ">u2", ">i8", ">u4")


@pytest.fixture(params=["oracle-double", "host-abi", pytest.param("hip", marks=pytest.mark.gpu)])
def tbackend(request, monkeypatch):
    if request.param == "oracle-double":
        from oracle import fake_device

        fake_device.install(monkeypatch)
    elif request.param == "host-abi":
        import host_abi_device

        host_abi_device.install(monkeypatch)
    return request.param


def _same(got, want):
    """bit-exact AND dtype-exact, result in native byte order like numpy's"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype.isnative
    assert got.dtype == want.dtype, f"dtype {got.dtype}, numpy {want.dtype}"
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def _field(shape, dtype, seed):
    """seeded values every tested dtype holds exactly, stored in `dtype` (byte order included)"""
    v = np.round(R.synthetic_field(shape, seed) * 200.0)
    dt = np.dtype(dtype)
    if dt.kind == "u":
        v = v + 100.0
    if dt.kind == "f":
        v = v / 8.0
    return v.astype(dt)


def _metric(shape, dtype, seed):
    return (np.round(R.synthetic_metric(shape, seed) / 16.0) / 4.0).astype(np.dtype(dtype))


def _setup(dtype, mdtype=None, nz=3, ny=6, nx=16):
    mdtype = mdtype or (dtype if np.dtype(dtype).kind == "f" else ">f8")
    T = _field((nz, ny, nx), dtype, 2)
    dx, dy, dz = _metric((ny, nx), mdtype, 31), _metric((ny, nx), mdtype, 32), _metric((nz,), mdtype, 33)
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) + 0.5), "Zl": ("Zl", np.arange(nz) * 1.0)}
    ds = Dataset({"T": (("Z", "YC", "XC"), T), "dxC": (("YC", "XG"), dx), "dxF": (("YC", "XC"), _metric((ny, nx), mdtype, 34)),
                  "dyC": (("YG", "XC"), dy), "drF": (("Z",), dz)}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"},
                metrics={("X",): ["dxC", "dxF"], ("Y",): ["dyC"], ("Z",): ["drF"]}, autoparse_metadata=False)
    return grid, ds, T, dx, dy, dz


# ----------------------------------------------------------------------------------------------
# the intake rule itself (host logic)
# ----------------------------------------------------------------------------------------------
def test_host_intake_never_reads_a_dtype_by_name():
    for code in NON_NATIVE + (">f2",):
        a = np.arange(7).astype(code)
        view, swap = DT.host_intake(a)
        assert swap == a.dtype.itemsize and view.dtype.isnative and view.dtype == a.dtype.newbyteorder("=")
        assert view.tobytes() == a.tobytes()                      # raw bytes: the swap happens after the copy
        assert np.array_equal(view.byteswap(), a)                 # ... and yields numpy's values
        assert DT.np_dtype(a) == a.dtype.newbyteorder("=") and DT.np_dtype(a).isnative
    for code in ("<f8", "f4", "i8", "u1", "?", ">i1", ">u1"):
        a = np.ones(3, dtype=code)
        view, swap = DT.host_intake(a)
        assert swap == 0 and view.dtype.isnative
    for bad in (np.complex64, np.complex128, "M8[ns]", "U3", object):
        with pytest.raises(TypeError, match="not supported by the MI355X backend"):
            DT.host_intake(np.zeros(3, dtype=bad))
    if np.dtype(np.longdouble).itemsize > 8:
        with pytest.raises(TypeError, match="not supported"):
            DT.host_intake(np.zeros(3, dtype=np.longdouble))


def test_metric_steps_follow_numpy_promotion():
    f2, f4, f8 = np.float16, np.float32, np.float64
    assert DT.metric_steps(f8, f8, f8) == (False, False) and DT.metric_steps(f4, f4, f4) == (False, False)
    assert DT.metric_steps(f4, None, f8) == (False, True)       # diff rounded to float32, then / float64
    assert DT.metric_steps(f4, f8, f8) == (False, False)        # the product is float64 already
    assert DT.metric_steps(f4, f4, f8) == (False, True)
    assert DT.metric_steps(f8, None, f4) == (False, False)
    assert DT.metric_steps(f2, f2, f2) == (True, True) and DT.metric_steps(f2, None, None) == (False, False)
    assert DT.metric_steps(f2, f4, f4) == (False, False) and DT.metric_steps(f2, None, f4) == (False, True)
    assert DT.metric_steps(">f4", None, ">f8") == (False, True)
    assert DT.half_result(f2) and DT.half_result(f2, np.int8) and not DT.half_result(f2, np.int16)
    assert not DT.half_result(f2, f4) and DT.float_of(f2) == f4 and DT.float_of(f2, np.int16) == f4


# ----------------------------------------------------------------------------------------------
# non-native fields AND metrics through the operators: bit- and dtype-equal to numpy
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", NON_NATIVE)
def test_non_native_arrays_through_the_grid_operators(tbackend, dtype):
    grid, ds, T, dx, dy, dz = _setup(dtype)
    assert not T.dtype.isnative and not dx.dtype.isnative
    _same(grid.diff(ds["T"], "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic"))
    _same(grid.diff(ds["T"], "Y").values, R.stencil1d("diff", T, 1, 1, 0, "extend"))
    _same(grid.interp(ds["T"], "X").values, R.stencil1d("interp", T, 2, 1, 0, "periodic"))
    _same(grid.interp(ds["T"], "Z", fill_value=2).values, R.stencil1d("interp", T, 0, 1, 0, "fill", 2))
    _same(grid.max(ds["T"], "Y").values, R.stencil1d("max", T, 1, 1, 0, "extend"))
    _same(grid.derivative(ds["T"], "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx[None]))
    _same(grid.derivative(ds["T"], "Y").values, R.stencil1d("diff", T, 1, 1, 0, "extend", m_out=dy[None]))
    _same(grid.diff(ds["T"], "X", metric_weighted=("X",)).values,
          R.stencil1d("diff", T, 2, 1, 0, "periodic", m_in=np.asarray(ds["dxF"].data)[None], m_out=dx[None]))
    _same(grid.cumsum(ds["T"], "Z").values, R.grid_cumsum(T, 0, "center", "left", "fill"))
    _same(grid.cumsum(ds["T"], "Y", to="left").values, R.grid_cumsum(T, 1, "center", "left", "extend"))
    _same(grid.integrate(ds["T"], "Z").values, R.integrate(T, 0, dz[:, None, None]))
    _same(grid.integrate(ds["T"], "Y").values, R.integrate(T, 1, _center_metric(grid, ds, "Y")))
    _same((ds["T"] * 2).values, T * 2)
    _same((ds["T"] * ds["drF"]).values, T * dz[:, None, None])


def _center_metric(grid, ds, axis):
    """the metric `integrate` finds for a centre field (interpolated from the one given at the left position)"""
    return np.asarray(grid.get_metric(ds["T"], (axis,)).values)[None]


@pytest.mark.parametrize("dtype", (">f4", ">f8", ">i4"))
def test_non_native_device_resident_and_mixed_residency(tbackend, dtype):
    """the field uploaded once (`to_device`), metrics still non-native host arrays: same results"""
    grid, ds, T, dx, dy, dz = _setup(dtype)
    dev = ds["T"].to_device() if tbackend == "hip" else ds["T"]
    _same(grid.derivative(dev, "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx[None]))
    _same(grid.cumsum(dev, "Z").values, R.grid_cumsum(T, 0, "center", "left", "fill"))


def test_mds_style_record_read_with_fromfile(tbackend, tmp_path):
    """an MDS `.data` record as MITgcm writes it -- big-endian float32, no header -- read with `np.fromfile(dtype=">f4")`
    and handed straight to `Grid.diff` / `Grid.interp` (what a user of the reference does with xmitgcm-less scripts)"""
    nz, ny, nx = 4, 8, 32
    native = (np.round(R.synthetic_field((nz, ny, nx), 7) * 4096.0) / 64.0).astype(np.float32)
    path = tmp_path / "T.0000000001.data"
    native.astype(">f4").tofile(path)
    rec = np.fromfile(path, dtype=">f4").reshape(nz, ny, nx)
    assert rec.dtype.str == ">f4" and np.array_equal(rec, native)
    grid, ds, *_ = _setup(">f4", nz=nz, ny=ny, nx=nx)
    da = DataArray(rec, ("Z", "YC", "XC"))
    _same(grid.diff(da, "X").values, R.stencil1d("diff", native, 2, 1, 0, "periodic"))
    _same(grid.interp(da, "Y").values, R.stencil1d("interp", native, 1, 1, 0, "extend"))
    # a read-only memory map of the same file (numpy.memmap keeps the dtype): same thing
    mm = np.memmap(path, dtype=">f4", mode="r", shape=(nz, ny, nx))
    _same(grid.diff(DataArray(mm, ("Z", "YC", "XC")), "X").values, R.stencil1d("diff", native, 2, 1, 0, "periodic"))


def test_non_native_strided_view(tbackend):
    """a transposed big-endian view (not contiguous): laid out, then swapped"""
    grid, ds, T, *_ = _setup(">f8")
    Tt = np.ascontiguousarray(T.transpose(1, 0, 2)).transpose(1, 0, 2)  # same values, F-like strides, still >f8
    assert not Tt.flags.c_contiguous and Tt.dtype.str == ">f8"
    _same(grid.diff(DataArray(Tt, ("Z", "YC", "XC")), "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic"))


# ----------------------------------------------------------------------------------------------
# mixed float32 / float64: each step in numpy's dtype (a float32 difference is rounded BEFORE a float64 metric divides)
# ----------------------------------------------------------------------------------------------
def test_float32_field_with_float64_metrics_rounds_where_numpy_rounds(tbackend):
    nz, ny, nx = 3, 6, 16
    T = R.synthetic_field((nz, ny, nx), 2).astype(np.float32)          # full-mantissa float32 values
    dx64 = R.synthetic_metric((ny, nx), 31)
    dz64 = R.synthetic_metric((nz,), 33)
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "Z": ("Z", np.arange(nz) + 0.5), "Zl": ("Zl", np.arange(nz) * 1.0)}
    ds = Dataset({"T": (("Z", "YC", "XC"), T), "dxC": (("YC", "XG"), dx64), "dxF": (("YC", "XC"), dx64 + 1.0),
                  "drF": (("Z",), dz64)}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Z": "fill"}, metrics={("X",): ["dxC", "dxF"], ("Z",): ["drF"]},
                autoparse_metadata=False)
    want = R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx64[None])     # float32 diff, then / float64
    assert want.dtype == np.float64
    assert not np.array_equal(want, R.stencil1d("diff", T.astype(np.float64), 2, 1, 0, "periodic", m_out=dx64[None]))
    _same(grid.derivative(ds["T"], "X").values, want)
    # metric_weighted: the float64 product comes first, everything after it is float64
    _same(grid.diff(ds["T"], "X", metric_weighted=("X",)).values,
          R.stencil1d("diff", T, 2, 1, 0, "periodic", m_in=(dx64 + 1.0)[None], m_out=dx64[None]))
    _same(grid.integrate(ds["T"], "Z").values, R.integrate(T, 0, dz64[:, None, None]))


# ----------------------------------------------------------------------------------------------
# float16: numpy's dtype; single operations bit for bit, sums within float16 rounding of numpy's own float16 sums
# ----------------------------------------------------------------------------------------------
def _half_field(shape, seed, order=">"):
    return (np.round(R.synthetic_field(shape, seed) * 2000.0) / 16.0).astype(np.dtype(np.float16).newbyteorder(order))


@pytest.mark.parametrize("order", ("=", ">"))
def test_float16_computes_like_numpy(tbackend, order):
    nz, ny, nx = 3, 6, 16
    T = _half_field((nz, ny, nx), 2, order)
    m16 = (np.round(R.synthetic_metric((ny, nx), 31) / 64.0) / 8.0 + 1.0).astype(np.float16)
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) + 0.5), "Zl": ("Zl", np.arange(nz) * 1.0)}
    ds = Dataset({"T": (("Z", "YC", "XC"), T), "dxC": (("YC", "XG"), m16), "dxF": (("YC", "XC"), m16 + np.float16(1))}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, metrics={("X",): ["dxC", "dxF"]},
                autoparse_metadata=False)
    for op in ("diff", "interp", "min", "max"):
        for axis, num, bc in (("X", 2, "periodic"), ("Y", 1, "extend"), ("Z", 0, "fill")):
            want = R.stencil1d(op, T, num, 1, 0, bc)
            assert want.dtype == np.float16
            _same(getattr(grid, op)(ds["T"], axis).values, want)
    _same(grid.derivative(ds["T"], "X").values, R.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=m16[None]))
    _same(grid.diff(ds["T"], "X", metric_weighted=("X",)).values,
          R.stencil1d("diff", T, 2, 1, 0, "periodic", m_in=(m16 + np.float16(1))[None], m_out=m16[None]))
    # two axes in one call: numpy rounds to float16 BETWEEN the axes, so the fused two-axis kernel (float32 throughout) is not
    # taken -- one axis at a time (a 16-cell row AND, on a second grid, an 18-cell one: not a multiple of the float32 lane vector)
    _same(grid.interp(ds["T"], ["X", "Y"]).values, R.stencil1d("interp", R.stencil1d("interp", T, 2, 1, 0, "periodic"), 1, 1, 0, "extend"))
    _same(grid.max(ds["T"], ["Y", "X"]).values, R.stencil1d("max", R.stencil1d("max", T, 1, 1, 0, "extend"), 2, 1, 0, "periodic"))
    T18 = _half_field((nz, ny, 18), 5, order)
    ds18 = Dataset({"T": (("Z", "YC", "XC"), T18)}, dict(coords, XC=("XC", np.arange(18) + 0.5), XG=("XG", np.arange(18) * 1.0)))
    g18 = Grid(ds18, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
               padding={"X": "periodic", "Y": "extend"}, autoparse_metadata=False)
    _same(g18.interp(ds18["T"], ["X", "Y"]).values, R.stencil1d("interp", R.stencil1d("interp", T18, 2, 1, 0, "periodic"), 1, 1, 0, "extend"))
    _same((ds["T"] * 0.5).values, T * 0.5)
    _same((ds["T"] / ds["dxF"]).values, T / (m16 + np.float16(1))[None])
    # float16 next to a float32 / float64 operand promotes like numpy
    _same((ds["T"] * DataArray(m16.astype(np.float32), ("YC", "XC"))).values, T * m16.astype(np.float32)[None])
    # prefix sums / sums: numpy rounds every partial sum to float16, the lanes here carry float32 partial sums -- the
    # result is float16 and within the float16 rounding numpy's own order of operations accumulates (n * eps * max|sum|)
    got = grid.cumsum(ds["T"], "Z").values
    want = R.grid_cumsum(T, 0, "center", "left", "fill")
    assert got.dtype == want.dtype == np.float16
    exact = R.grid_cumsum(T.astype(np.float64), 0, "center", "left", "fill")
    tol = nz * 2.0**-10 * np.abs(exact).max() + 2.0**-10
    assert np.abs(got.astype(np.float64) - exact).max() <= tol and np.abs(want.astype(np.float64) - exact).max() <= tol


def test_float16_interp_overflow_is_the_documented_deviation(tbackend):
    """`(a + b) / 2.0` in float16 overflows where a + b > 65504 (numpy: inf); the float32 lanes return the finite mean"""
    a = np.array([[60000.0, 60000.0, 1.0, 3.0]], dtype=np.float16)
    ds = Dataset({"a": (("Y", "XC"), a)}, {"XC": ("XC", np.arange(4) + 0.5), "XG": ("XG", np.arange(4) * 1.0), "Y": ("Y", [0.0])})
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}}, padding="extend", autoparse_metadata=False)
    got = grid.interp(ds["a"], "X").values
    with np.errstate(over="ignore"):
        want = R.stencil1d("interp", a, 1, 1, 0, "extend")
    assert got.dtype == np.float16 and np.array_equal(got[0, 2:], want[0, 2:])
    if tbackend == "oracle-double":
        assert np.isinf(want[0, 0]) and np.isinf(got[0, 0])      # numpy itself
    else:
        assert np.isinf(want[0, 0]) and got[0, 0] == np.float16(60000.0)


# ----------------------------------------------------------------------------------------------
# dtypes nobody serves: a TypeError that names them, on every backend
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bad", (np.complex64, np.complex128))
def test_complex_is_refused_with_a_clear_error(tbackend, bad):
    grid, ds, T, *_ = _setup(">f8")
    z = DataArray(np.asarray(T, dtype=np.float64).astype(bad), ("Z", "YC", "XC"))
    for call in (lambda: grid.diff(z, "X"), lambda: grid.cumsum(z, "Z"), lambda: grid.integrate(z, "Z"), lambda: z * 2.0):
        with pytest.raises(TypeError, match="complex arrays are not supported by the MI355X backend"):
            call()


# ----------------------------------------------------------------------------------------------
# GPU only: the ABI pieces the intake uses
# ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_bswap_and_half_conversion_on_the_gpu():
    import torch

    from xgcm_amd import _hip, device as dev

    rng = np.random.default_rng(5)
    for code, n in ((">i2", 1003), (">u2", 8), (">f2", 77), (">f4", 1001), (">i8", 513)):
        a = rng.integers(-30000, 30000, n).astype(code)
        t = dev._raw_device(a)
        assert DT.np_dtype(t) == a.dtype.newbyteorder("=") and np.array_equal(t.cpu().numpy(), a)
    # float16 <-> float32 / float64 conversions are numpy's astype, every float16 bit pattern
    bits = np.arange(65536, dtype=np.uint16)
    h = bits.view(np.float16)
    for wide in (np.float32, np.float64):
        up = dev.tohost(dev.convert(torch.from_numpy(h.copy()).cuda(), wide))
        assert up.dtype == wide and np.array_equal(up, h.astype(wide), equal_nan=True)
    with np.errstate(over="ignore"):
        for wide in (np.float32, np.float64):
            x = np.concatenate([h.astype(wide), (rng.standard_normal(200000) * 10.0 ** rng.integers(-9, 6, 200000)).astype(wide),
                                np.array([65504.0, 65519.9, 65520.0, 65536.0, 2.0**-24, 2.0**-25, 2.0**-25 * 1.0001, 5.96e-8,
                                          -2.0**-25, 6.1e-5, 6.10352e-5], dtype=wide)])
            down = dev.tohost(dev.convert(torch.from_numpy(x).cuda(), np.float16))
            assert down.dtype == np.float16 and np.array_equal(down.view(np.uint16)[~np.isnan(x)], x.astype(np.float16).view(np.uint16)[~np.isnan(x)])
            assert np.isnan(down[np.isnan(x)]).all()
    assert _hip.load().xg_bswap(None, 0, 2, None) == 0


@pytest.mark.parametrize("wdtype", (np.float16, np.int8, np.bool_))
def test_float16_average_divides_by_the_sum_of_the_weights(tbackend, wdtype):
    """`Grid.average` of a float16 field with a float16 / int8 / bool metric is sum(x * w) / sum(w) over the
    valid cells -- the float16 branch of `device.reduce1d` once folded the weight into the field and left the kernel a plain COUNT
    for its denominator (sum(x * w) / N).  One dim (the one-pass mean mode) and two dims (the pair mode)."""
    nz, ny, nx = 3, 6, 16
    T = _half_field((nz, ny, nx), 9, "=")
    T[1, 2, 3] = np.nan
    if wdtype is np.bool_:
        w = (R.synthetic_field((ny, nx), 11) > -0.25)
        w[0, 0] = True
    else:
        w = (np.round(R.synthetic_metric((ny, nx), 31) / 400.0) + 3.0).astype(wdtype)  # 5 ... 8: far from 1
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) + 0.5)}
    ds = Dataset({"T": (("Z", "YC", "XC"), T), "dxF": (("YC", "XC"), w), "rA": (("YC", "XC"), w)}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z"}},
                padding={"X": "periodic", "Y": "extend"}, metrics={("X",): ["dxF"], ("X", "Y"): ["rA"]}, autoparse_metadata=False)
    x64, w64 = T.astype(np.float64), w.astype(np.float64)[None]
    valid = ~np.isnan(x64)
    for dims, red in ((["X"], (2,)), (["X", "Y"], (1, 2))):
        got = grid.average(ds["T"], dims).values
        want = np.where(valid, x64 * w64, 0.0).sum(red) / np.where(valid, w64, 0.0).sum(red)
        wrong = np.where(valid, x64 * w64, 0.0).sum(red) / valid.sum(red)  # what the folded weight computed
        assert got.dtype == np.float16 and got.shape == want.shape
        tol = 8 * nx * ny * 2.0**-10 * np.abs(want).max() + 2.0**-9
        assert np.abs(got.astype(np.float64) - want).max() <= tol
        if wdtype is not np.bool_:
            assert np.abs(wrong - want).max() > 4 * tol  # (the test can tell the two apart)
