"""GPU parity at BASELINE.json's FULL sizes through size-independent properties of the domain
(the oracle cannot finish a 648 M-cell field in seconds): exact round trips and identities on
integer-valued data (every sum/difference is exact in float64, so equality is bitwise), plus
spot slabs of the full-size outputs compared with the CPU oracle.

torch is used here only to GENERATE integer-valued inputs and to COMPARE results (checker code);
everything under test runs through xgcm_amd.Grid -> C ABI -> HIP kernels.
"""

import numpy as np
import pytest

from oracle import refimpl as R

pytestmark = pytest.mark.gpu

import os

# BASELINE.json's sizes; XG_FULLSIZE_SHAPE="z,y,x,n5" shrinks them only to dry-run the test LOGIC on the CPU double
NZ, NY, NX = (int(v) for v in os.environ.get("XG_FULLSIZE_SHAPE", "75,2400,3600,4320").split(",")[:3])
N5 = int(os.environ.get("XG_FULLSIZE_SHAPE", "75,2400,3600,4320").split(",")[3])


@pytest.fixture(scope="module")
def env():
    import torch

    from xgcm_amd import DataArray, Dataset, Grid
    from xgcm_amd import device as D

    coords = {"XC": ("XC", np.arange(NX) + 0.5), "XG": ("XG", np.arange(NX) * 1.0),
              "YC": ("YC", np.arange(NY) + 0.5), "YG": ("YG", np.arange(NY) * 1.0),
              "Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0), "Zp1": ("Zp1", np.arange(NZ + 1) * 1.0)}
    ones = DataArray(D.synthetic((NZ,), 0, 0, 0.0, 1.0), ("Z",))
    two_d = DataArray(D.synthetic((NY, NX), 0, 0, 0.0, 4.0), ("YC", "XG"))  # constant 4.0: exact division
    ds = Dataset({"drF": ones, "dxC": two_d}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl", "outer": "Zp1"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"},
                metrics={("Z",): ["drF"], ("X",): ["dxC"]}, autoparse_metadata=False)

    def ints(seed, shape=(NZ, NY, NX), dims=("Z", "YC", "XC")):
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        t = torch.randint(-(2 ** 20), 2 ** 20, shape, device="cuda", dtype=torch.int32, generator=g).to(torch.float64)
        return DataArray(t, dims)

    return {"torch": torch, "grid": grid, "ints": ints, "D": D, "DataArray": DataArray, "Dataset": Dataset, "Grid": Grid}


def _same(torch, a, b):
    return bool(torch.equal(a, b))


def test_cumsum_then_diff_round_trip_full_size(env):
    """decode(encode(x)) == x: diff(outer->center) of cumsum(center->outer, fill 0) is the identity."""
    torch, grid = env["torch"], env["grid"]
    x = env["ints"](1)
    c = grid.cumsum(x, "Z", to="outer")
    assert c.dims == ("Zp1", "YC", "XC") and c.shape == (NZ + 1, NY, NX)
    back = grid.diff(c, "Z", to="center")
    assert back.dims == x.dims and _same(torch, back.data, x.data)
    # the reversed scan lands on the left faces; its round trip needs the sign flipped
    cr = grid.cumsum(x, "Z", to="outer", reverse=True)
    back = grid.diff(cr, "Z", to="center")
    assert _same(torch, (back * -1.0).data, x.data)
    # integrate == last plane of the natural cumsum (unit metric)
    nat = grid.cumsum(x, "Z", to="left", reverse=True)  # natural for reverse: total sits at index 0
    tot = grid.integrate(x, "Z")
    assert tot.dims == ("YC", "XC") and _same(torch, tot.data, nat.data[0])
    assert _same(torch, tot.data, c.data[NZ])


@pytest.mark.parametrize("dtype", ["int64", "int32", "uint32"])
def test_integer_fields_full_size(env, dtype):
    """integer lanes at BASELINE size (*_i64 / *_i32 builds): values over the whole range of the dtype, so every sum and
    difference WRAPS -- and the round trips still close exactly, because two's-complement arithmetic is a ring:
    diff(outer->center) of cumsum(center->outer) is the identity modulo 2^64, a periodic diff sums to zero along its axis,
    min <= max in the dtype's own order; spot slabs against numpy."""
    torch, grid, DataArray = env["torch"], env["grid"], env["DataArray"]
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    raw = torch.randint(-(2 ** 63), 2 ** 63 - 1, (NZ, NY, NX), device="cuda", dtype=torch.int64, generator=g)
    tdt = getattr(torch, dtype)
    x = raw if dtype == "int64" else raw.to(torch.int32).view(tdt)
    da = DataArray(x, ("Z", "YC", "XC"))
    c = grid.cumsum(da, "Z", to="outer")
    want_c = {"int64": torch.int64, "int32": torch.int64, "uint32": torch.uint64}[dtype]
    assert c.data.dtype == want_c and c.shape == (NZ + 1, NY, NX)
    back = grid.diff(c, "Z")
    wide = x.to(torch.int64) if dtype != "uint32" else x.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    assert _same(torch, back.data.view(torch.int64), wide)
    for ax, dim in (("X", 2), ("Y", 1)):
        d = grid.diff(da, ax, padding="periodic")
        assert d.data.dtype == tdt
        lanes = d.data if dtype == "int64" else d.data.view(torch.int32)
        total = lanes.to(torch.int64).sum(dim)  # wraps modulo 2^64; the true sum of a periodic difference is 0 modulo 2^bits
        mod = total if dtype == "int64" else total & 0xFFFFFFFF
        assert not bool(mod.any())
        lo, hi = grid.min(da, ax, padding="periodic").data, grid.max(da, ax, padding="periodic").data
        if dtype == "uint32":
            lo, hi = lo.view(torch.int32).to(torch.int64) & 0xFFFFFFFF, hi.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        assert bool((lo <= hi).all())
    k = min(2, NZ - 1)
    slab = x[k:k + 1, 100 % NY:100 % NY + 3].cpu().numpy() if dtype != "uint32" else x.view(torch.int32)[k:k + 1, 100 % NY:100 % NY + 3].cpu().numpy().view(np.uint32)
    for op in ("diff", "max", "interp"):
        got = getattr(grid, op)(da, "X", padding="periodic").data[k:k + 1, 100 % NY:100 % NY + 3]
        got = got.view(torch.int32).cpu().numpy().view(np.uint32) if got.dtype == getattr(torch, "uint32", None) else got.cpu().numpy()
        np.testing.assert_array_equal(got, R.stencil1d(op, slab, 2, 1, 0, "periodic"))


def test_chained_scan_along_y_full_size(env):
    """cumsum along Y of the full 75 x 2400 x 3600 field runs as the chained flat launch (K5c: 75 chunks of 32 rows per
    column hand their running sum on): on integer-valued data every partial sum is exact, so (a) the last kept row equals
    the plain sum along Y, (b) differencing the scan gives the field back, (c) the marching kernel (scan_chain=0) agrees
    bit for bit, forward and reversed; (d) spot columns against the numpy oracle."""
    from xgcm_amd import _hip

    torch, grid, D = env["torch"], env["grid"], env["D"]
    x = env["ints"](5)
    c = grid.cumsum(x, "Y", to="left", padding="fill")  # out[0] = 0, out[j] = sum of rows < j
    assert c.dims == ("Z", "YG", "XC")
    back = D.stencil1d("diff", c.data, 1, 0, 1, "fill", 0.0)  # c[j + 1] - c[j], last one against the fill
    assert _same(torch, back[:, :-1], x.data[:, :-1])
    total = D.reduce1d(x.data, 1, None, False)
    incl = D.cumsum1d(x.data, 1, 0, 0, 0, 0, None)  # plain inclusive scan
    assert _same(torch, incl[:, -1], total)
    keep = _hip.get_tunable("scan_chain")
    try:
        for rev in (False, True):
            _hip.set_tunable("scan_chain", 1)
            chained = D.cumsum1d(x.data, 1, 0, 1, 1, 0, "extend", 0.0, rev, True)
            _hip.set_tunable("scan_chain", 0)
            marched = D.cumsum1d(x.data, 1, 0, 1, 1, 0, "extend", 0.0, rev, True)
            assert _same(torch, chained, marched)
    finally:
        _hip.set_tunable("scan_chain", keep)
    for z, xs in ((0, 0), (NZ // 2, NX // 2 - 1), (NZ - 1, NX - 8)):  # spot columns (8 wide) against the oracle
        col = D.tohost(x.data[z, :, xs:xs + 8])
        np.testing.assert_array_equal(D.tohost(incl[z, :, xs:xs + 8]), np.cumsum(col, axis=0))


def test_transforms_row_staging_full_size(env):
    """75 -> 50 levels on the full 2400 x 3600 columns, random-walk theta (the lanes of a wave emit a target level at
    different source levels): outputs through the LDS ring / the per-wave accumulator window == direct stores / a
    window per lane, bit for bit; spot columns against numpy.interp and the oracle's conservative loop."""
    from oracle import transform as TR
    from xgcm_amd import _hip

    torch, D = env["torch"], env["D"]
    m = 50
    phi = D.synthetic((NZ, NY, NX), 41)
    theta = torch.cumsum(D.synthetic((NZ, NY, NX), 42, 0, 2.0, 0.01), 0)
    theta_o = torch.cat([theta[:1] - 1.0, theta], 0)
    levels = torch.linspace(1.0, 0.9 * NZ, m, dtype=torch.float64, device="cuda").reshape(m, 1, 1)
    edges = torch.linspace(0.0, 1.05 * NZ, m + 1, dtype=torch.float64, device="cuda")
    keep = {k: _hip.get_tunable(k) for k in ("transform_stage", "transform_win")}
    try:
        _hip.set_tunable("transform_stage", 0)
        _hip.set_tunable("transform_win", 2)
        lin0 = D.transform_linear(phi, theta, levels, 0)
        con0 = D.transform_conservative(phi, theta_o, edges, 0)
        _hip.set_tunable("transform_stage", keep["transform_stage"])
        _hip.set_tunable("transform_win", keep["transform_win"])
        lin1 = D.transform_linear(phi, theta, levels, 0)
        con1 = D.transform_conservative(phi, theta_o, edges, 0)
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)
    assert lin1.shape == (m, NY, NX) and bool(torch.equal(torch.nan_to_num(lin0, nan=-7.0), torch.nan_to_num(lin1, nan=-7.0)))
    assert bool(torch.equal(torch.nan_to_num(con0, nan=-7.0), torch.nan_to_num(con1, nan=-7.0)))
    for y, x in ((0, 0), (NY // 2, NX // 3), (NY - 1, NX - 4)):
        ph, th, tho = (D.tohost(a[:, y, x:x + 4]).T for a in (phi, theta, theta_o))  # (4 columns, levels)
        want = TR.interp_1d_linear(ph, th, D.tohost(levels).ravel(), mask_edges=True)
        np.testing.assert_array_equal(D.tohost(lin1[:, y, x:x + 4]).T, want)
        np.testing.assert_array_equal(D.tohost(con1[:, y, x:x + 4]).T, TR.interp_1d_conservative(ph, tho, D.tohost(edges)))


def test_multi_axis_integrals_full_size(env):
    """integrate / average over [X, Y] and [X, Y, Z] with separable metrics (area(Y, X) x thickness(Z)) at full size: on
    integer data with power-of-two metrics every partial sum is exact whatever the order, so the staged reductions
    (factor by factor; numerator and denominator side by side for the mean) must reproduce torch's plain sums."""
    torch, D, DataArray, Dataset, Grid = env["torch"], env["D"], env["DataArray"], env["Dataset"], env["Grid"]
    coords = {"XC": ("XC", np.arange(NX) + 0.5), "XG": ("XG", np.arange(NX) * 1.0), "YC": ("YC", np.arange(NY) + 0.5),
              "YG": ("YG", np.arange(NY) * 1.0), "Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0)}
    area = DataArray(D.synthetic((NY, NX), 0, 0, 0.0, 2.0), ("YC", "XC"))
    thick = DataArray(D.synthetic((NZ,), 0, 0, 0.0, 4.0), ("Z",))
    grid = Grid(Dataset({"rA": area, "drF": thick}, coords),
                coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding="fill", metrics={("X", "Y"): ["rA"], ("Z",): ["drF"]}, autoparse_metadata=False)
    x = env["ints"](9)
    per_level = x.data.sum(dim=(1, 2))
    got = grid.integrate(x, ["X", "Y"])
    assert got.dims == ("Z",) and _same(torch, got.data, per_level * 2.0)
    vol = grid.integrate(x, ["X", "Y", "Z"])
    assert vol.dims == () and float(vol.data) == float(x.data.sum() * 8.0)
    avg = grid.average(x, ["X", "Y"])  # (tensor / tensor: torch turns a division by a Python scalar into a product with its reciprocal)
    assert _same(torch, avg.data, (per_level * 2.0) / torch.full_like(per_level, 2.0 * NY * NX))
    avg3 = grid.average(x, ["X", "Y", "Z"])
    assert float(avg3.data) == float(x.data.sum() * 8.0) / (8.0 * NZ * NY * NX)


def test_two_axis_metric_weighted_full_size(env):
    """xg_stencil2d_metric at full size (band-major order, four levels per task sharing three random metric planes, the
    last group of 75 = 18 x 4 + 3 short) == the two metric-carrying 1-D launches, both axis orders, bit for bit."""
    torch, D = env["torch"], env["D"]
    T = D.synthetic((NZ, NY, NX), 61)
    m1, m2, m3 = (D.synthetic((NY, NX), s, 0, 1000.0, 1000.0) for s in (62, 63, 64))
    for order in (0, 1):
        fused = D.stencil2d("interp", T, order, (1, 0), "periodic", 0.0, (1, 0), "extend", 0.0, metrics=(m1, m2, m3))
        first, second = ((2, "periodic"), (1, "extend")) if order == 0 else ((1, "extend"), (2, "periodic"))
        step = D.stencil1d("interp", T, first[0], 1, 0, first[1], 0.0, m_in=m1[None], m_out=m2[None])
        seq = D.stencil1d("interp", step, second[0], 1, 0, second[1], 0.0, m_in=m2[None], m_out=m3[None])
        assert _same(torch, fused, seq)


def test_linearity_and_shift_invariance_full_size(env):
    torch, grid = env["torch"], env["grid"]
    a, b = env["ints"](2), env["ints"](3)
    s = a + b
    for fn in ("diff", "interp"):
        for ax in ("X", "Y"):
            f = getattr(grid, fn)
            lhs = f(a, ax) + f(b, ax)
            assert _same(torch, lhs.data, f(s, ax).data), (fn, ax)
    # periodic X: the operator commutes with a cyclic shift of the field
    rolled = env["DataArray"](torch.roll(a.data, 7, dims=2).contiguous(), a.dims)
    assert _same(torch, grid.diff(rolled, "X").data, torch.roll(grid.diff(a, "X").data, 7, dims=2))
    # min/max: min + max == left + right  (exact), and min <= interp <= max
    mn, mx = grid.min(a, "Y"), grid.max(a, "Y")
    assert _same(torch, (mn + mx).data, (grid.interp(a, "Y") * 2.0).data)
    assert bool((mn.data <= mx.data).all())
    # derivative with a constant metric 4.0 is an exact scaling of diff
    assert _same(torch, (grid.derivative(a, "X") * 4.0).data, grid.diff(a, "X").data)


def test_curl_of_gradient_is_zero_config5_size(env):
    """zeta(grad(phi)) == 0 exactly on the 4320 x 4320 x 90 C-grid (fused kernel, periodic)."""
    torch = env["torch"]
    nz, n = 90, 4320
    coords = {"XC": ("XC", np.arange(n) + 0.5), "XG": ("XG", np.arange(n) * 1.0),
              "YC": ("YC", np.arange(n) + 0.5), "YG": ("YG", np.arange(n) * 1.0), "Z": ("Z", np.arange(nz) * 1.0)}
    grid = env["Grid"](env["Dataset"](coords=coords),
                       coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                       padding="periodic", autoparse_metadata=False)
    phi = env["ints"](5, (nz, n, n), ("Z", "YC", "XC"))
    u = grid.diff(phi, "X")  # (Z, YC, XG)
    v = grid.diff(phi, "Y")  # (Z, YG, XC)
    zeta = grid.vorticity(u, v, metric_weighted=False)
    assert zeta.dims == ("Z", "YG", "XG")
    assert int(torch.count_nonzero(zeta.data)) == 0
    # and a non-gradient field gives the same bits fused and unfused
    w = env["ints"](6, (nz, n, n), ("Z", "YG", "XC"))
    fused = grid.vorticity(u, w, metric_weighted=False)
    chain = grid.diff(w, "X") - grid.diff(u, "Y")
    assert _same(torch, fused.data, chain.data)


def test_full_size_outputs_match_oracle_on_spot_slabs(env):
    """random (real-valued) field: edge and interior slabs of the full-size result vs the oracle."""
    grid, D = env["grid"], env["D"]
    T = env["DataArray"](D.synthetic((NZ, NY, NX), 2), ("Z", "YC", "XC"))
    host = lambda t: t.cpu().numpy()  # noqa: E731
    dX, iY, dZ = grid.diff(T, "X"), grid.interp(T, "Y"), grid.derivative(T, "X")
    cZ = grid.cumsum(T, "Z")
    for z in (0, 37, NZ - 1):
        slab = host(T.data[z])  # (NY, NX)
        assert np.array_equal(host(dX.data[z]), R.stencil1d("diff", slab, 1, 1, 0, "periodic"))
        assert np.array_equal(host(iY.data[z]), R.stencil1d("interp", slab, 0, 1, 0, "extend"))
        assert np.array_equal(host(dZ.data[z]), R.stencil1d("diff", slab, 1, 1, 0, "periodic", m_out=np.full((NY, NX), 4.0)))
    cols = host(T.data[:, 1200:1204, :])  # (NZ, 4, NX) column block
    assert np.array_equal(host(cZ.data[:, 1200:1204, :]), R.grid_cumsum(cols, 0, "center", "left", "fill"))


def test_more_than_2_to_32_cells_in_one_call(env):
    """5.2 G cells (41 GB) in ONE call: 64-bit addressing and the host-side launch splitting
    (per-launch item counts stay < 2^31).  The last record must equal the same record alone."""
    torch, grid, D = env["torch"], env["grid"], env["D"]
    nt = 8
    T4 = env["DataArray"](D.synthetic((nt, NZ, NY, NX), 4), ("time", "Z", "YC", "XC"))
    assert T4.data.numel() > 2 ** 32
    last = env["DataArray"](T4.data[nt - 1].contiguous(), ("Z", "YC", "XC"))
    for fn, ax, kw in (("diff", "X", {}), ("interp", "Y", {}), ("diff", "Z", {}), ("cumsum", "Z", {}),
                       ("derivative", "X", {})):
        full = getattr(grid, fn)(T4, ax, **kw)
        one = getattr(grid, fn)(last, ax, **kw)
        assert full.dims[0] == "time" and _same(torch, full.data[nt - 1], one.data), (fn, ax)
        del full, one
    tot = grid.integrate(T4, "Z")
    assert tot.dims == ("time", "YC", "XC") and _same(torch, tot.data[nt - 1], grid.integrate(last, "Z").data)


# ----------------------------------------------------------------------------------------------
# Config 3 / 4 / 5 at full size with RANDOM metrics against oracle slabs: the
# z-banded 2-D-metric paths re-order rows, so a wrong (z, y) -> metric offset must not be able to hide
# behind a constant metric.  Reference contract: xgcm/test/test_metrics_ops.py:59-64,134-216 (bitwise).
# ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def env3(env):
    D, DataArray = env["D"], env["DataArray"]
    coords = {"XC": ("XC", np.arange(NX) + 0.5), "XG": ("XG", np.arange(NX) * 1.0),
              "YC": ("YC", np.arange(NY) + 0.5), "YG": ("YG", np.arange(NY) * 1.0),
              "Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0)}
    met = lambda shape, seed: D.synthetic(shape, seed, 0, 1000.0, 1000.0)  # noqa: E731  1000 * (1 + u) > 0
    dv = {"dxC": DataArray(met((NY, NX), 31), ("YC", "XG")), "dyC": DataArray(met((NY, NX), 32), ("YG", "XC")),
          "dxT": DataArray(met((NY, NX), 35), ("YC", "XC")), "dyT": DataArray(met((NY, NX), 36), ("YC", "XC")),
          "drF": DataArray(met((NZ,), 33), ("Z",)), "drC": DataArray(met((NZ,), 34), ("Zl",))}
    grid = env["Grid"](env["Dataset"](dv, coords),
                       coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                               "Z": {"center": "Z", "left": "Zl"}},
                       padding={"X": "periodic", "Y": "extend", "Z": "fill"},
                       metrics={("X",): ["dxC", "dxT"], ("Y",): ["dyC", "dyT"], ("Z",): ["drF", "drC"]},
                       autoparse_metadata=False)
    T = DataArray(D.synthetic((NZ, NY, NX), 2), ("Z", "YC", "XC"))
    host = {k: R.synthetic_metric((NY, NX), s) for k, s in (("dxC", 31), ("dyC", 32), ("dxT", 35), ("dyT", 36))}
    host["drF"], host["drC"] = R.synthetic_metric((NZ,), 33), R.synthetic_metric((NZ,), 34)
    return {"grid": grid, "T": T, "m": host}


LEVELS = (0, NZ // 2, NZ - 1)


def _h(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)


def test_config3_derivative_with_random_2d_metrics_full_size(env3):
    """derivative(T,'X') / dxC(YC,XG) and derivative(T,'Y') / dyC(YG,XC): whole levels 0 / 37 / 74 bit for bit"""
    grid, T, m = env3["grid"], env3["T"], env3["m"]
    dX, dY = grid.derivative(T, "X"), grid.derivative(T, "Y")
    assert dX.dims == ("Z", "YC", "XG") and dY.dims == ("Z", "YG", "XC")
    for z in LEVELS:
        slab = _h(T.data[z])
        assert np.array_equal(_h(dX.data[z]), R.stencil1d("diff", slab, 1, 1, 0, "periodic", m_out=m["dxC"])), z
        assert np.array_equal(_h(dY.data[z]), R.stencil1d("diff", slab, 0, 1, 0, "extend", m_out=m["dyC"])), z


def test_config3_metric_weighted_interp_full_size(env3):
    """interp(T, ax, metric_weighted=ax): (T * m_in) padded, interpolated, / m_out -- two broadcast 2-D metrics"""
    grid, T, m = env3["grid"], env3["T"], env3["m"]
    iX = grid.interp(T, "X", metric_weighted="X")
    iY = grid.interp(T, "Y", metric_weighted="Y")
    for z in LEVELS:
        slab = _h(T.data[z])
        assert np.array_equal(_h(iX.data[z]), R.stencil1d("interp", slab, 1, 1, 0, "periodic", m_in=m["dxT"], m_out=m["dxC"])), z
        assert np.array_equal(_h(iY.data[z]), R.stencil1d("interp", slab, 0, 1, 0, "extend", m_in=m["dyT"], m_out=m["dyC"])), z


def test_config3_vertical_derivative_and_integrals_full_size(env3, env):
    """derivative(T,'Z') / drC(Zl) (1-D), integrate(T,'Z') * drF(Z) (1-D) and the integral of a field on the
    w-levels weighted by a 3-D drC(Zl,YC,XC): column blocks at the first, a middle and the last rows"""
    grid, T, m = env3["grid"], env3["T"], env3["m"]
    D, DataArray = env["D"], env["DataArray"]
    dZ, iZ = grid.derivative(T, "Z"), grid.integrate(T, "Z")
    assert dZ.dims == ("Zl", "YC", "XC") and iZ.dims == ("YC", "XC")
    blocks = (slice(0, 3), slice(NY // 2 - 1, NY // 2 + 2), slice(NY - 3, NY))
    for b in blocks:
        cols = _h(T.data[:, b, :])
        assert np.array_equal(_h(dZ.data[:, b, :]), R.stencil1d("diff", cols, 0, 1, 0, "fill", 0.0, m_out=m["drC"][:, None, None]))
        assert np.array_equal(_h(iZ.data[b, :]), R.integrate(cols, 0, m["drF"][:, None, None]))
    # 3-D cell thickness (partial cells): a grid whose Z metric at the w-levels is a full (Zl, YC, XC) array
    drC3 = DataArray(D.synthetic((NZ, NY, NX), 37, 0, 1000.0, 1000.0), ("Zl", "YC", "XC"))
    g3 = env["Grid"](env["Dataset"]({"drC3": drC3}, {"Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0)}),
                     coords={"Z": {"center": "Z", "left": "Zl"}}, padding="fill", metrics={("Z",): ["drC3"]},
                     autoparse_metadata=False)
    W = DataArray(T.data, ("Zl", "YC", "XC"))
    i3 = g3.integrate(W, "Z")
    c3 = g3.cumint(W, "Z")  # to the centres; fill halo
    for b in blocks:
        cols, w = _h(T.data[:, b, :]), _h(drC3.data[:, b, :])
        assert np.array_equal(_h(i3.data[b, :]), R.integrate(cols, 0, w))
        assert np.array_equal(_h(c3.data[:, b, :]), R.grid_cumsum(cols, 0, "left", "center", "fill", m_in=w))


def test_config3_integrals_along_the_contiguous_axis_full_size(env3):
    """integrate / average along X at full size -- the workgroup-per-row kernels (K4w; two levels share the row's dxT
    vectors) and the plain `sum`: whole levels 0 / 37 / 74 against numpy's own sums of the same products.  The kernels
    re-associate (numpy is pairwise along the last axis): 1e-12 relative to the sum of the |terms|, NaN cells skipped."""
    grid, T, m = env3["grid"], env3["T"], env3["m"]
    iX, aX, sX = grid.integrate(T, "X"), grid.average(T, "X"), T.sum("XC")
    assert iX.dims == aX.dims == sX.dims == ("Z", "YC")
    for z in LEVELS:
        slab = _h(T.data[z])
        w = m["dxT"]
        scale = np.abs(slab * w).sum(axis=1)
        np.testing.assert_allclose(_h(iX.data[z]), (slab * w).sum(axis=1), rtol=0, atol=1e-12 * scale.max())
        np.testing.assert_allclose(_h(aX.data[z]), (slab * w).sum(axis=1) / w.sum(axis=1), rtol=0, atol=1e-12 * (scale / w.sum(axis=1)).max())
        np.testing.assert_allclose(_h(sX.data[z]), slab.sum(axis=1), rtol=0, atol=1e-12 * np.abs(slab).sum(axis=1).max())


def test_config4_cumsum_center_to_outer_full_size(env3):
    """cumsum(T,'Z') center->left and center->outer (fill) against oracle column blocks (config 4's two ops)"""
    T = env3["T"]
    g = env3["grid"]
    from xgcm_amd import Dataset, Grid

    grid = Grid(Dataset(coords={"Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0), "Zp1": ("Zp1", np.arange(NZ + 1) * 1.0)}),
                coords={"Z": {"center": "Z", "left": "Zl", "outer": "Zp1"}}, padding="fill", autoparse_metadata=False)
    cl, co = grid.cumsum(T, "Z"), grid.cumsum(T, "Z", to="outer")
    assert cl.dims == ("Zl", "YC", "XC") and co.dims == ("Zp1", "YC", "XC") and co.shape[0] == NZ + 1
    for b in (slice(0, 2), slice(NY // 2, NY // 2 + 4), slice(NY - 2, NY)):
        cols = _h(T.data[:, b, :])
        assert np.array_equal(_h(cl.data[:, b, :]), R.grid_cumsum(cols, 0, "center", "left", "fill"))
        assert np.array_equal(_h(co.data[:, b, :]), R.grid_cumsum(cols, 0, "center", "outer", "fill"))
    assert g is not None


def test_config5_vorticity_fill_with_random_area_full_size(env):
    """(diff(V,'X') - diff(U,'Y')) / rAz on 4320 x 4320 x 90 with `fill` and a RANDOM rAz(YG,XG): whole levels
    0 / 45 / 89 (row 0 and column 0 carry the fill halo) against the oracle chain, fused and unfused"""
    torch, D, DataArray = env["torch"], env["D"], env["DataArray"]
    nz, n = (90, N5) if N5 == 4320 else (6, N5)
    coords = {"XC": ("XC", np.arange(n) + 0.5), "XG": ("XG", np.arange(n) * 1.0),
              "YC": ("YC", np.arange(n) + 0.5), "YG": ("YG", np.arange(n) * 1.0)}
    ds = env["Dataset"]({"rAz": DataArray(D.synthetic((n, n), 53, 0, 1000.0, 1000.0), ("YG", "XG"))}, coords)
    grid = env["Grid"](ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}},
                       padding="fill", metrics={("X", "Y"): ["rAz"]}, autoparse_metadata=False)
    U = DataArray(D.synthetic((nz, n, n), 51), ("Z", "YC", "XG"))
    V = DataArray(D.synthetic((nz, n, n), 52), ("Z", "YG", "XC"))
    zeta = grid.vorticity(U, V)
    assert zeta.dims == ("Z", "YG", "XG")
    area = R.synthetic_metric((n, n), 53)
    for z in (0, nz // 2, nz - 1):
        want = R.vorticity(_h(U.data[z]), _h(V.data[z]), area, "fill", "fill")
        got = _h(zeta.data[z])
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[:, 0], want[:, 0])  # the fill halo row / column
        assert np.array_equal(got, want), z
    chain = (grid.diff(V, "X") - grid.diff(U, "Y")) / ds["rAz"].reset_coords(drop=True)
    assert np.array_equal(_h(zeta.data), _h(chain.data)) if not hasattr(zeta.data, "is_cuda") else bool(torch.equal(zeta.data, chain.data))
