"""GPU parity tests proper: every C-ABI compute entry point vs the CPU oracle on the same seeded
inputs.  Bit-exact (np.array_equal, NaN == NaN) wherever the summation order is defined;
1e-12 relative only for the contiguous-axis scan/reduce (re-associated sums)."""

import itertools

import numpy as np
import pytest

from oracle import refimpl as R

pytestmark = pytest.mark.gpu

PADS = [(1, 0), (0, 1), (1, 1), (0, 0)]
BCS = ["periodic", "fill", "extend"]
OPS = ["diff", "interp", "min", "max"]
# shapes chosen to hit: vector path (even inner), scalar path (odd inner), 1-D, 4-D, tiny axes
SHAPES = [(6, 10, 128), (3, 7, 33), (2, 5, 4, 258), (257,), (4, 2), (3, 130), (5, 1, 66)]


@pytest.fixture(scope="module")
def dev():
    from xgcm_amd import device

    return device


def _eq(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), f"max abs diff {np.nanmax(np.abs(a - b))}"


def _field(shape, seed, nan=False):
    a = R.synthetic_field(shape, seed)
    if nan and a.size > 3:
        a.reshape(-1)[[1, a.size // 2, a.size - 1]] = np.nan
    return a


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("op", OPS)
def test_stencil_all_axes_pads_bcs(dev, shape, op):
    a = _field(shape, 11, nan=op in ("min", "max"))
    for axis in range(len(shape)):
        for (lo, hi), bc in itertools.product(PADS, BCS):
            if shape[axis] + lo + hi - 1 < 1:
                continue
            exp = R.stencil1d(op, a, axis, lo, hi, bc, 1.25)
            got = dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, 1.25))
            _eq(got, exp)


def test_stencil_no_halo_needs_no_bc(dev):
    a = _field((4, 9, 20), 3)
    for axis in range(3):
        _eq(dev.tohost(dev.stencil1d("diff", a, axis, 0, 0, None)), R.stencil1d("diff", a, axis, 0, 0, None))


def _metric_for(shape, keep_dims, seed):
    """metric broadcasting over all dims except `keep_dims` (full extent there)."""
    mshape = [s if d in keep_dims else 1 for d, s in enumerate(shape)]
    return R.synthetic_metric(mshape, seed)


@pytest.mark.parametrize("shape", [(5, 6, 64), (3, 7, 33), (2, 3, 4, 130), (3, 37, 130), (1, 20, 256)])
@pytest.mark.parametrize("op", ["diff", "interp"])
def test_stencil_metric_weighted_bitwise(dev, shape, op):
    """metric_weighted == (x*m_in) -> op -> / m_out bitwise (reference test_metrics_ops.py:59-64)."""
    nd = len(shape)
    a = _field(shape, 5)
    for axis in range(nd):
        for (lo, hi), bc in itertools.product([(1, 0), (0, 1), (1, 1), (0, 0)], BCS):
            n_out = shape[axis] + lo + hi - 1
            if n_out < 1:
                continue
            oshape = list(shape)
            oshape[axis] = n_out
            # a few broadcast patterns: horizontal 2-D metric, 1-D along axis, full N-D
            patterns = [set(range(nd)), {axis}, {nd - 1, max(nd - 2, 0)}]
            for keep in patterns:
                m_in = _metric_for(shape, keep, 31)
                m_out = _metric_for(oshape, keep, 32)
                exp = R.stencil1d(op, a, axis, lo, hi, bc, 0.75, m_in=m_in, m_out=m_out)
                got = dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, 0.75, m_in=m_in, m_out=m_out))
                _eq(got, exp)
                # derivative: m_out only
                exp = R.stencil1d(op, a, axis, lo, hi, bc, 0.75, m_out=m_out)
                got = dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, 0.75, m_out=m_out))
                _eq(got, exp)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_contig_metric_stencil_many_levels(dev, dtype):
    """K1r, z-shared: `derivative` / `metric_weighted` along X of (Z, Y, X) with metrics shared by MANY levels (the other
    stencil tests have 2-6): level counts that are not a multiple of the levels per wave-task, ragged x-tiles, every pad and
    boundary condition -- the oracle's bits."""
    if True:
        for shape in ((37, 5, 130), (33, 3, 256), (19, 4, 64), (70, 2, 66)):
            a = _field(shape, 47).astype(dtype)
            for (lo, hi), bc in itertools.product([(1, 0), (0, 1)], BCS):
                for keep_dims in ({1, 2}, {2}):
                    m_in = _metric_for(shape, keep_dims, 48).astype(dtype)
                    m_out = _metric_for(shape, keep_dims, 49).astype(dtype)
                    for op in ("diff", "interp"):
                        for kw in ({"m_out": m_out}, {"m_in": m_in, "m_out": m_out}, {"m_in": m_in}):
                            exp = R.stencil1d(op, a, 2, lo, hi, bc, 0.75, **kw)
                            _eq(dev.tohost(dev.stencil1d(op, a, 2, lo, hi, bc, 0.75, **kw)), exp)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ys", [0, 12, 14, 18, 22])
def test_strided_metric_stencil_y_stacked_workgroups(dev, dtype, ys):
    """K2Sm: `derivative` / `metric_weighted` along Y of (Z, Y, X) with metrics shared by the levels -- the 4 waves of a
    workgroup are 4 consecutive segments of one x-tile and hand their top row (already multiplied by its metric, or
    replaced by the fill value) to the wave above through LDS.  Every task shape, row counts that are not a multiple of the
    workgroup's rows, level counts that are not a multiple of the levels per wave, ragged x-tiles, every pad and boundary
    condition -- the oracle's bits, as with K2S (ys = 0)."""
    from xgcm_amd import _hip
    keep = {k: _hip.get_tunable(k) for k in ("met_ys1", "met_ys2")}
    _hip.set_tunable("met_ys1", ys)
    _hip.set_tunable("met_ys2", ys)
    try:
        for shape in ((5, 37, 130), (3, 64, 128), (9, 33, 66), (2, 8, 256), (7, 9, 64)):
            a = _field(shape, 41).astype(dtype)
            for (lo, hi), bc in itertools.product([(1, 0), (0, 1), (1, 1), (0, 0)], BCS):
                n_out = shape[1] + lo + hi - 1
                oshape = (shape[0], n_out, shape[2])
                for keep_dims in ({1, 2}, {1}):
                    m_in = _metric_for(shape, keep_dims, 43).astype(dtype)
                    m_out = _metric_for(oshape, keep_dims, 44).astype(dtype)
                    for op in ("diff", "interp"):
                        for kw in ({"m_out": m_out}, {"m_in": m_in, "m_out": m_out}, {"m_in": m_in}):
                            exp = R.stencil1d(op, a, 1, lo, hi, bc, 0.75, **kw)
                            _eq(dev.tohost(dev.stencil1d(op, a, 1, lo, hi, bc, 0.75, **kw)), exp)
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ys", [0, 1])
def test_two_axis_metric_kernels_every_form(dev, dtype, ys):
    """`xg_stencil2d_metric` (metric_weighted on both axes) through both of its kernels -- K8 (met_ys 0) and the y-stacked
    K8y (1): every operator, both pads on both axes, every boundary pair incl. non-zero fills, level counts that are not a
    multiple of the four a task carries, row lengths around multiples of the wave, heights that are not a multiple of the
    workgroup's rows -- equal to the two metric-carrying 1-D passes of the oracle, bit for bit.  (Round 4 also ran a third
    form through this test, a march along Y with overlapped x-tiles: bit-exact, no faster, removed -- profiles/EXPERIMENTS.md.)"""
    from xgcm_amd import _hip
    keep = _hip.get_tunable("met_ys")
    _hip.set_tunable("met_ys", ys)
    nv = 2 if dtype == np.float64 else 4
    try:
        for nz, ny, nvec in ((9, 37, 130), (6, 16, 63 * 2), (5, 19, 63 * 2 + 1), (4, 8, 64), (7, 33, 63 * 3 - 1)):
            shape = (nz, ny, nvec * nv)
            a = _field(shape, 75).astype(dtype)
            m1, m2, m3 = (R.synthetic_metric((1,) + shape[1:], 76 + k).astype(dtype) for k in range(3))
            for op in OPS:
                for (padx, pady), (bcx, bcy) in itertools.product(itertools.product([(1, 0), (0, 1)], [(1, 0), (0, 1)]),
                                                                 [("periodic", "extend"), ("fill", "periodic"), ("extend", "fill")]):
                    t = R.stencil1d(op, a, 2, *padx, bcx, dtype(0.75), m1, m2)
                    want = R.stencil1d(op, t, 1, *pady, bcy, dtype(-1.5), m2, m3)
                    got = dev.tohost(dev.stencil2d(op, a, 0, padx, bcx, 0.75, pady, bcy, -1.5, metrics=(m1[0], m2[0], m3[0])))
                    _eq(got, want)
    finally:
        _hip.set_tunable("met_ys", keep)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ys", [0, 12, 14, 22])
def test_strided_metric_stencil_with_per_face_metrics(dev, dtype, ys):
    """Round 4 (DESIGN rule 17): a (Z, face, Y, X) field -- MITgcm's LLC / cubed-sphere layout -- whose metrics (face, Y, X)
    change from face to face and are shared by the levels only.  Two outer dims, the metric broadcast along the slower one:
    the band-major launches (K2S, K2Sm) now run over (face, row) under the levels instead of falling back to the unbanded
    order (derivative Y on the cubed sphere: 0.57, metric_weighted Y 0.45 of 8 TB/s, profiles/history/r04d_f2.log).  Level counts
    that are not a multiple of the levels per task, one face, many faces, every pad / boundary / metric combination, a leading
    record dim on top: the oracle's bits."""
    from xgcm_amd import _hip
    keep = {k: _hip.get_tunable(k) for k in ("met_ys1", "met_ys2")}
    _hip.set_tunable("met_ys1", ys)
    _hip.set_tunable("met_ys2", ys)
    try:
        for shape in ((5, 3, 37, 130), (7, 6, 33, 128), (2, 1, 40, 64), (3, 2, 4, 9, 256)):
            a = _field(shape, 47).astype(dtype)
            ax = len(shape) - 2
            for (lo, hi), bc in itertools.product([(1, 0), (0, 1), (1, 1), (0, 0)], BCS):
                oshape = list(shape)
                oshape[ax] = shape[ax] + lo + hi - 1
                mshape_in = (1,) * (len(shape) - 3) + tuple(shape[-3:])
                mshape_out = (1,) * (len(shape) - 3) + tuple(oshape[-3:])
                m_in = R.synthetic_metric(mshape_in, 48).astype(dtype)
                m_out = R.synthetic_metric(mshape_out, 49).astype(dtype)
                for op in ("diff", "interp"):
                    for kw in ({"m_out": m_out}, {"m_in": m_in, "m_out": m_out}, {"m_in": m_in}):
                        exp = R.stencil1d(op, a, ax, lo, hi, bc, 0.75, **kw)
                        _eq(dev.tohost(dev.stencil1d(op, a, ax, lo, hi, bc, 0.75, **kw)), exp)
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


@pytest.mark.parametrize("shape", [(6, 10, 128), (3, 7, 33), (2, 5, 4, 66), (300,), (3, 700), (2, 2050), (3, 1024), (9, 4100)])
def test_cumsum_all(dev, shape):
    a = _field(shape, 7, nan=True)
    nd = len(shape)
    for axis in range(nd):
        contiguous_axis = axis == nd - 1
        for reverse, skipna in itertools.product([False, True], [True, False]):
            for tl, th, pl, ph in [(0, 0, 0, 0), (0, 1, 1, 0), (0, 1, 0, 0), (0, 0, 1, 0), (1, 0, 0, 1), (1, 0, 0, 0),
                                   (0, 0, 0, 1), (1, 1, 1, 1)]:
                if shape[axis] - tl - th < 1:
                    continue
                for bc in BCS:
                    exp = R.cumsum1d(a, axis, tl, th, pl, ph, bc, 0.5, reverse, skipna)
                    got = dev.tohost(dev.cumsum1d(a, axis, tl, th, pl, ph, bc, 0.5, reverse, skipna))
                    if contiguous_axis and shape[axis] > 1:
                        assert got.shape == exp.shape
                        np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-12, equal_nan=True)
                    else:
                        _eq(got, exp)


def test_cumsum_long_rows_metric(dev):
    """wave-per-row scan kernel (even rows >= 256) with metrics, both directions."""
    shape = (3, 5, 2048)
    a = _field(shape, 10)
    m_in = _metric_for(shape, {1, 2}, 43)
    for reverse in (False, True):
        tl, th, pl, ph = (0, 1, 1, 0) if not reverse else (1, 0, 0, 1)
        m_out = _metric_for(shape, {2}, 44)
        for bc in BCS:
            exp = R.cumsum1d(a, 2, tl, th, pl, ph, bc, 0.25, reverse, True, m_in, m_out)
            got = dev.tohost(dev.cumsum1d(a, 2, tl, th, pl, ph, bc, 0.25, reverse, True, m_in, m_out))
            np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-11)  # values ~1e1, a few cross zero


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_long_strided_march_with_few_columns(dev, dtype):
    """n >= 256 along a strided axis and few wave-tasks: the scans / reductions switch to one element
    per lane with 16 loads in flight; results stay bit-identical to the sequential numpy order."""
    a = _field((3, 300, 128), 88, nan=True).astype(dtype)
    w = R.synthetic_metric((1, 300, 1), 89).astype(dtype)
    for rev in (False, True):
        exp = R.cumsum1d(a, 1, 0, 1, 1, 0, "fill", dtype(0.5), rev, True)
        _eq(dev.tohost(dev.cumsum1d(a, 1, 0, 1, 1, 0, "fill", 0.5, rev, True)), exp)
    _eq(dev.tohost(dev.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, False, False, w, None)), R.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, False, False, w, None))
    for skipna in (True, False):
        exp = R.integrate(a, 1, w, skipna)
        _eq(dev.tohost(dev.reduce1d(a, 1, w, skipna)), exp.astype(dtype) if exp.dtype != dtype else exp)
    # metrics that vary along the lanes too (float32 marches with 8-byte lanes: two metric values per lane)
    w2 = R.synthetic_metric((1, 300, 128), 90).astype(dtype)
    m3 = R.synthetic_metric((3, 300, 128), 91).astype(dtype)
    _eq(dev.tohost(dev.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, False, True, w2, m3)), R.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, False, True, w2, m3))
    _eq(dev.tohost(dev.reduce1d(a, 1, w2, True)), R.integrate(a, 1, w2, True).astype(dtype))
    _eq(dev.tohost(dev.reduce1d(a, 1, w2, "mean_valid")),
        (dev.tohost(dev.reduce1d(a, 1, w2, True)) / dev.tohost(dev.reduce1d(a, 1, w2, "valid"))).astype(dtype))
    # weights that do not depend on the outer index: four levels per wave share each weight load (K4z); 9 levels =
    # two full groups and a short one, 4-D outer dims, every mode
    b = _field((3, 3, 260, 64), 92, nan=True).astype(dtype)
    wb = R.synthetic_metric((1, 1, 260, 64), 93).astype(dtype)
    for skipna in (True, False):
        _eq(dev.tohost(dev.reduce1d(b, 2, wb, skipna)), R.integrate(b, 2, wb, skipna).astype(dtype))
    ones = (~np.isnan(b)).astype(dtype)
    _eq(dev.tohost(dev.reduce1d(b, 2, wb, "valid")), R.integrate(ones, 2, wb, False).astype(dtype))
    with np.errstate(invalid="ignore", divide="ignore"):
        _eq(dev.tohost(dev.reduce1d(b, 2, wb, "mean_valid")),
            (R.integrate(b, 2, wb, True) / R.integrate(ones, 2, wb, False)).astype(dtype))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cumsum_chained_chunks(dev, dtype):
    """K5c: the long strided-axis scan as a chained flat launch (chunks of 32 / 16 rows handing their running sum to
    the next chunk of the column).  Forced on for short columns too (scan_chain=2): every trim / pad / boundary /
    direction / NaN mode, ragged last chunks, metrics, several columns per XCD band -- bit-identical to the
    sequential numpy order, and twice in a row (the workspace must come back clean)."""
    from xgcm_amd import _hip
    before, before_zl, before_lds = _hip.get_tunable("scan_chain"), _hip.get_tunable("reduce_zl"), _hip.get_tunable("reduce_ldsw")
    _hip.set_tunable("scan_chain", 2)
    _hip.set_tunable("reduce_ldsw", 0)  # (the LDS-weight march would take the long weighted reductions: this test is about the chain)
    try:
        for shape in ((3, 300, 128), (2, 70, 130), (5, 64, 66), (1, 33, 2), (9, 97, 700)):
            a = _field(shape, 7, nan=True).astype(dtype)
            for reverse, skipna in itertools.product([False, True], [True, False]):
                for tl, th, pl, ph in [(0, 0, 0, 0), (0, 1, 1, 0), (1, 0, 0, 1), (0, 0, 1, 0), (0, 0, 0, 1), (1, 1, 1, 1)]:
                    for bc in BCS:
                        exp = R.cumsum1d(a, 1, tl, th, pl, ph, bc, dtype(0.5), reverse, skipna)
                        for _ in range(2):
                            _eq(dev.tohost(dev.cumsum1d(a, 1, tl, th, pl, ph, bc, 0.5, reverse, skipna)), exp)
            m_in = R.synthetic_metric((1,) + shape[1:], 43).astype(dtype)
            m_out = R.synthetic_metric(shape, 44).astype(dtype)
            for m_i, m_o in ((m_in, None), (None, m_out), (m_in, m_out)):
                exp = R.cumsum1d(a, 1, 0, 1, 1, 0, "extend", dtype(0.0), False, True, m_i, m_o)
                _eq(dev.tohost(dev.cumsum1d(a, 1, 0, 1, 1, 0, "extend", 0.0, False, True, m_i, m_o)), exp)
        # the leading axis of a 2-D array (one outer index, many tiles) and a 4-D array (outer dims coalesce)
        b = _field((130, 520), 8).astype(dtype)
        _eq(dev.tohost(dev.cumsum1d(b, 0, 0, 0, 0, 0, None, 0.0, False, True)), R.cumsum1d(b, 0, 0, 0, 0, 0, None, dtype(0), False, True))
        c4 = _field((2, 3, 80, 64), 9, nan=True).astype(dtype)
        _eq(dev.tohost(dev.cumsum1d(c4, 2, 0, 1, 1, 0, "periodic", 0.0, True, True)), R.cumsum1d(c4, 2, 0, 1, 1, 0, "periodic", dtype(0), True, True))
        # K4c, the weighted reductions in the same form: every mode against the marching kernel (scan_chain=0), which
        # the tests above pin to the oracle -- same additions in the same order, so the same bits
        for shape, wshape in (((3, 300, 128), (1, 300, 128)), ((2, 70, 130), (2, 70, 130)), ((4, 97, 66), (1, 97, 1)), ((2, 3, 80, 64), (1, 1, 80, 64))):
            a = _field(shape, 17, nan=True).astype(dtype)
            w = R.synthetic_metric(wshape, 18).astype(dtype)
            axis = len(shape) - 2
            for mode in (True, False, "valid", "all", "mean_valid", "mean_all", "pair_valid", "pair_all"):
                _hip.set_tunable("scan_chain", 0)
                ref = dev.tohost(dev.reduce1d(a, axis, w, mode))
                _hip.set_tunable("scan_chain", 2)
                for zl in (1, 2, 4):  # levels per task sharing the weight rows (K4c / K4cz)
                    _hip.set_tunable("reduce_zl", zl)
                    for _ in range(2):
                        _eq(dev.tohost(dev.reduce1d(a, axis, w, mode)), ref)
    finally:
        _hip.set_tunable("scan_chain", before)
        _hip.set_tunable("reduce_zl", before_zl)
        _hip.set_tunable("reduce_ldsw", before_lds)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_weighted_reduction_with_weights_through_lds(dev, dtype):
    """K4L: `integrate` / `average` along a long strided axis with weights shared by the outer indices -- the 4 waves of a
    workgroup march 4 consecutive levels of one x-tile, the weight rows of a block of 32 rows go through LDS once per
    workgroup.  Every mode, columns that are not a multiple of the block, outer extents that are not a multiple of 4,
    ragged x-tiles, NaNs -- against the oracle (sequential order: bit-exact) and against the marching kernel."""
    from xgcm_amd import _hip
    keep = {k: _hip.get_tunable(k) for k in ("reduce_ldsw", "scan_chain", "reduce_ldsw_u", "march_ofast", "reduce_zmarch")}
    try:
        _hip.set_tunable("reduce_zmarch", 0)
        for shape, wshape in (((3, 300, 128), (1, 300, 128)), ((6, 64, 130), (1, 64, 130)), ((5, 97, 66), (1, 97, 66)),
                              ((2, 3, 80, 64), (1, 1, 80, 64)), ((9, 1000, 70), (1, 1000, 70)), ((4, 65, 2), (1, 65, 2))):
            a = _field(shape, 27, nan=True).astype(dtype)
            w = R.synthetic_metric(wshape, 28).astype(dtype)
            axis = len(shape) - 2
            for mode in (True, False, "valid", "all", "mean_valid", "mean_all", "pair_valid", "pair_all"):
                _hip.set_tunable("reduce_ldsw", 0)
                _hip.set_tunable("scan_chain", 0)
                _hip.set_tunable("march_ofast", 0)  # the plain weighted march, x-tiles / outer indices fastest
                ref = dev.tohost(dev.reduce1d(a, axis, w, mode))
                _hip.set_tunable("march_ofast", 1)
                _eq(dev.tohost(dev.reduce1d(a, axis, w, mode)), ref)
                _hip.set_tunable("scan_chain", keep["scan_chain"])
                for order in (1, 2):  # x-tiles fastest / level groups fastest in the work order
                    _hip.set_tunable("reduce_ldsw", order)
                    for lu in (8, 16):  # rows per block
                        _hip.set_tunable("reduce_ldsw_u", lu)
                        _eq(dev.tohost(dev.reduce1d(a, axis, w, mode)), ref)
                for zm in (1208, 1216, 1308, 1312, 1316, 1408):  # K4Z (round 6): ZL levels per wave share the weight row in registers (+ 1000: float32 too)
                    _hip.set_tunable("reduce_zmarch", zm)
                    _eq(dev.tohost(dev.reduce1d(a, axis, w, mode)), ref)
                _hip.set_tunable("reduce_zmarch", keep["reduce_zmarch"])
            with np.errstate(invalid="ignore"):
                _eq(dev.tohost(dev.reduce1d(a, axis, w, True)), R.integrate(a, axis, np.broadcast_to(w, shape), True).astype(dtype))
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_metric_views_unaligned_and_strided(dev, dtype):
    """The marches along a strided axis decide the FORM of their metric loads once per wave (`met_vec_all`: one aligned vector
    per lane, or element by element).  A metric handed over as a VIEW -- starting one element into its allocation with an odd
    row pitch, or stored transposed (element step != 1 along the lanes) -- must take the element-wise form and give the bits
    of the same metric laid out contiguously: weighted sums / means along Y (K4Z), along Z with a (Y, X) weight, and the
    weighted scans along Y (chained) and Z (march)."""
    import torch

    shape = (5, 130, 132)
    a = _field(shape, 41, nan=True).astype(dtype)
    w = R.synthetic_metric((1, 130, 132), 42).astype(dtype)
    t = dev.asdevice(a)
    wc = dev.asdevice(w)
    pitch = torch.zeros((1, 130, 133), dtype=wc.dtype, device=wc.device)
    pitch[:, :, 1:] = wc
    transposed = wc.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert transposed.stride(2) == 130 and pitch[:, :, 1:].storage_offset() == 1
    for view in (pitch[:, :, 1:], transposed):
        for mode in (True, False, "valid", "all", "mean_valid", "mean_all", "pair_valid", "pair_all"):
            _eq(dev.tohost(dev.reduce1d(t, 1, view, mode)), dev.tohost(dev.reduce1d(t, 1, wc, mode)))
            _eq(dev.tohost(dev.reduce1d(t, 0, view, mode)), dev.tohost(dev.reduce1d(t, 0, wc, mode)))
        for axis in (0, 1):
            for rev in (False, True):
                _eq(dev.tohost(dev.cumsum1d(t, axis, 0, 1, 1, 0, "fill", 0.0, rev, True, view, None)),
                    dev.tohost(dev.cumsum1d(t, axis, 0, 1, 1, 0, "fill", 0.0, rev, True, wc, None)))
    with np.errstate(invalid="ignore"):
        _eq(dev.tohost(dev.reduce1d(t, 1, pitch[:, :, 1:], True)), R.integrate(a, 1, np.broadcast_to(w, shape), True).astype(dtype))


def test_weighted_march_orders_under_every_banding_setting(dev):
    """`march_ofast` orders the plain weighted march outer-indices-fastest through the banded wave id,
    which walks ceil(grid / 8) workgroups per XCD band -- the grid must be rounded to a multiple of 8 for it also when
    `march_band` is off (it was not: task ids below the count stayed unvisited, part of the output unwritten).  Shapes whose
    workgroup count is NOT a multiple of 8, every combination of the two tunables, against the oracle (poisoned output)."""
    import torch

    from xgcm_amd import _hip
    keep = {k: _hip.get_tunable(k) for k in ("reduce_ldsw", "scan_chain", "march_ofast", "march_band")}
    try:
        _hip.set_tunable("reduce_ldsw", 0)
        _hip.set_tunable("scan_chain", 0)
        for shape, wshape in (((3, 40, 130), (1, 40, 130)), ((5, 33, 66), (1, 33, 66)), ((7, 20, 1000), (1, 20, 1000)),
                              ((2, 3, 17, 640), (1, 1, 17, 640)), ((9, 300, 70), (1, 300, 70))):
            a = _field(shape, 71, nan=True)
            w = R.synthetic_metric(wshape, 72)
            axis = len(shape) - 2
            with np.errstate(invalid="ignore"):
                want = R.integrate(a, axis, np.broadcast_to(w, shape), True)
            for band in (0, 1, 2):
                for ofast in (0, 1):
                    _hip.set_tunable("march_band", band)
                    _hip.set_tunable("march_ofast", ofast)
                    # poison the allocator's next block so that an unwritten cell cannot pass by luck
                    junk = torch.full(want.shape, float("nan"), dtype=torch.float64, device="cuda")
                    del junk
                    _eq(dev.tohost(dev.reduce1d(a, axis, w, True)), want)
                    got = dev.tohost(dev.cumsum1d(a, axis, 0, 0, 0, 0, None, 0.0, False, True))
                    _eq(got, R.cumsum1d(a, axis, 0, 0, 0, 0, None, 0.0, False, True))
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


def test_chained_scan_orders_with_a_metric_shared_by_the_levels(dev):
    """`scan_chain_tmaj = k`: the chained cumint / weighted reduction whose metric is shared by the outer indices numbers
    its columns x-tile-major and advances all levels of k x-tiles side by side (a traffic / time trade).
    Every order hands the same running sums on: bit-identical results for k = 0 (level-major bands) ... 6, column counts that
    do and do not fill the sub-bands, forward and reversed."""
    from xgcm_amd import _hip
    keep = {k: _hip.get_tunable(k) for k in ("scan_chain_tmaj", "reduce_ldsw")}
    try:
        _hip.set_tunable("reduce_ldsw", 0)  # the chained weighted reduction, not K4L
        for shape in ((7, 300, 200), (3, 130, 1100), (11, 96, 64)):
            a = _field(shape, 91, nan=True)
            w = R.synthetic_metric((1,) + shape[1:], 92)
            m_out = R.synthetic_metric((1, shape[1], shape[2]), 93)
            want_c = R.cumsum1d(a, 1, 0, 1, 1, 0, "fill", 0.0, False, True, np.broadcast_to(w, shape), m_out)
            want_r = R.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, True, True, np.broadcast_to(w, shape), None)
            with np.errstate(invalid="ignore"):
                want_s = R.integrate(a, 1, np.broadcast_to(w, shape), True)
            for k in (0, 1, 2, 4, 6):
                _hip.set_tunable("scan_chain_tmaj", k)
                _eq(dev.tohost(dev.cumsum1d(a, 1, 0, 1, 1, 0, "fill", 0.0, False, True, w, m_out)), want_c)
                _eq(dev.tohost(dev.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, True, True, w, None)), want_r)
                _eq(dev.tohost(dev.reduce1d(a, 1, w, True)), want_s)
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


def test_chained_scans_on_two_streams(dev):
    """The chained kernels keep their hand-off slots and ticket counters in a workspace PER STREAM: two streams running
    long scans / weighted reductions at the same time do not see each other's running sums."""
    import torch

    shape = (6, 1200, 256)
    a = dev.asdevice(_field(shape, 31, nan=True))
    b = dev.asdevice(_field(shape, 32))
    w = dev.asdevice(R.synthetic_metric((1,) + shape[1:], 33))
    want = [dev.tohost(dev.cumsum1d(a, 1, 0, 1, 1, 0, "extend", 0.0, False, True)), dev.tohost(dev.reduce1d(b, 1, w, "mean_valid")),
            dev.tohost(dev.cumsum1d(b, 1, 1, 0, 0, 1, "periodic", 0.0, True, True, w, None))]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    got = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            g0 = dev.cumsum1d(a, 1, 0, 1, 1, 0, "extend", 0.0, False, True)
        with torch.cuda.stream(s2):
            g1 = dev.reduce1d(b, 1, w, "mean_valid")
            g2 = dev.cumsum1d(b, 1, 1, 0, 0, 1, "periodic", 0.0, True, True, w, None)
        got.append((g0, g1, g2))
    torch.cuda.synchronize()
    for g in got:
        for x, y in zip(g, want):
            _eq(dev.tohost(x), y)


def test_cumsum_metric(dev):
    shape = (4, 9, 6, 34)
    a = _field(shape, 9)
    for axis in (1, 2, 3):
        for reverse in (False, True):
            tl, th, pl, ph = (0, 1, 1, 0) if not reverse else (1, 0, 0, 1)
            oshape = list(shape)
            m_in = _metric_for(shape, {axis}, 41)
            m_out = _metric_for(oshape, {2, 3}, 42)
            exp = R.cumsum1d(a, axis, tl, th, pl, ph, "fill", 0.0, reverse, True, m_in, m_out)
            got = dev.tohost(dev.cumsum1d(a, axis, tl, th, pl, ph, "fill", 0.0, reverse, True, m_in, m_out))
            if axis == 3:
                np.testing.assert_allclose(got, exp, rtol=1e-12, atol=0)
            else:
                _eq(got, exp)


# (rows of 512 cells or more take the workgroup-per-row kernels along the last axis -- K4w, and K4wz with its odd last level
# when the weights are shared by the levels: 5 and 3 levels here)
@pytest.mark.parametrize("shape", [(6, 10, 128), (3, 7, 33), (2, 5, 4, 66), (300,), (75, 6, 10), (5, 6, 1024), (3, 7, 1538),
                                   (2, 2, 4096)])
def test_reduce(dev, shape):
    a = _field(shape, 13, nan=True)
    nd = len(shape)
    for axis in range(nd):
        for skipna in (True, False):
            for keep in (None, {axis}, set(range(nd))):
                w = None if keep is None else _metric_for(shape, keep, 21)
                exp = R.integrate(a, axis, w, skipna)
                got = dev.tohost(dev.reduce1d(a, axis, w, skipna))
                if axis == nd - 1:
                    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-13, equal_nan=True)
                else:
                    _eq(got, exp)
        # the two denominators of a weighted mean: weights of the valid cells / of all cells, same pass
        w = _metric_for(shape, set(range(nd)), 22)
        for mode, ones in (("valid", (~np.isnan(a)).astype(a.dtype)), ("all", np.ones_like(a))):
            for ww in (w, None):
                exp = R.integrate(ones, axis, ww, False)
                got = dev.tohost(dev.reduce1d(a, axis, ww, mode))
                if axis == nd - 1:
                    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-13)
                else:
                    _eq(got, exp)
        # the weighted mean in ONE pass (numerator and denominator together, divided in the kernel) == the two
        # separate sums divided afterwards, bit for bit on every axis (same sums, same order, one IEEE division)
        with np.errstate(invalid="ignore", divide="ignore"):
            for mode, nmode, dmode in (("mean_valid", True, "valid"), ("mean_all", False, "all")):
                for ww in (w, None):
                    two_pass = dev.tohost(dev.reduce1d(a, axis, ww, nmode)) / dev.tohost(dev.reduce1d(a, axis, ww, dmode))
                    _eq(dev.tohost(dev.reduce1d(a, axis, ww, mode)), two_pass)
                    both = dev.tohost(dev.reduce1d(a, axis, ww, mode.replace("mean", "pair")))  # the two sums side by side
                    assert both.shape == (2,) + two_pass.shape
                    _eq(both[0], dev.tohost(dev.reduce1d(a, axis, ww, nmode)))
                    _eq(both[1], dev.tohost(dev.reduce1d(a, axis, ww, dmode)))
                    if axis != nd - 1:
                        ones = (~np.isnan(a)).astype(a.dtype) if dmode == "valid" else np.ones_like(a)
                        _eq(two_pass, R.integrate(a, axis, ww, nmode) / R.integrate(ones, axis, ww, False))


def test_pad_matches_numpy_chain(dev):
    a = _field((4, 5, 6), 17)
    cases = [
        ({2: (1, 1)}, {2: "periodic"}, {}),
        ({1: (0, 1)}, {1: "extend"}, {}),
        ({2: (0, 1), 1: (2, 0)}, {2: "periodic", 1: "fill"}, {1: np.nan}),
        ({1: (2, 0), 2: (0, 1)}, {2: "fill", 1: "periodic"}, {2: 1.5}),
        ({0: (7, 9), 2: (13, 3)}, {0: "periodic", 2: "extend"}, {}),
        ({0: (1, 2), 1: (2, 1), 2: (3, 3)}, {0: "fill", 1: "extend", 2: "periodic"}, {0: -2.0}),
    ]
    for widths, bc, fill in cases:
        exp = R.pad_nd(a, widths, bc, fill)
        got = dev.tohost(dev.pad_nd(a, widths, bc, fill))
        _eq(got, exp)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("tpw", [1, 2, 4])
def test_pad_long_rows_tiles_per_wave(dev, dtype, tpw):
    """the row-wise pad kernel (rows of >= 64 cells): 1 / 2 / 4 tiles of 64 lanes per wave-task, rows that are not a multiple
    of a tile or of a wave-task, padded and untouched innermost dim, every boundary mode, odd row lengths (rows that do not
    start on a 16-byte boundary) -- numpy's pad chain, bit for bit; and the elementwise operator with and without its 32-bit
    index decomposition on the same arrays"""
    from xgcm_amd import _hip
    keep = {k: _hip.get_tunable(k) for k in ("pad_tpw", "bin_idx32")}
    _hip.set_tunable("pad_tpw", tpw)
    try:
        for shape in ((3, 5, 70), (2, 4, 130), (2, 3, 300), (2, 2, 515), (1, 2, 1026)):
            a = _field(shape, 61).astype(dtype)
            cases = [
                ({2: (1, 1)}, {2: "periodic"}, {}),
                ({2: (2, 0)}, {2: "fill"}, {2: 1.5}),
                ({2: (0, 3)}, {2: "extend"}, {}),
                ({1: (1, 2)}, {1: "extend"}, {}),
                ({0: (1, 0), 1: (0, 1)}, {0: "fill", 1: "periodic"}, {0: -2.0}),
                ({0: (1, 1), 1: (2, 1), 2: (3, 2)}, {0: "periodic", 1: "fill", 2: "extend"}, {1: 0.25}),
                ({2: (70, 5)}, {2: "periodic"}, {}),
            ]
            for widths, bc, fill in cases:
                _eq(dev.tohost(dev.pad_nd(a, widths, bc, fill)), R.pad_nd(a, widths, bc, fill))
            b = R.synthetic_metric((1,) + shape[1:], 62).astype(dtype)
            for idx32 in (0, 1):
                _hip.set_tunable("bin_idx32", idx32)
                for op in ("mul", "div"):
                    _eq(dev.tohost(dev.binary(op, a, b)), R.binary(op, a, b))
    finally:
        for k, v in keep.items():
            _hip.set_tunable(k, v)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gather_random_token_maps(dev, dtype):
    """xg_gather vs the numpy decode of the token map (oracle/topology.py::gather_tokens): random
    sources, signs, fill slots, a partner with permuted dims, mapped dims not trailing."""
    from oracle import topology as T

    rng = np.random.default_rng(7)
    cases = [
        # (in shape, mapped flags, lo, hi, partner shape or None, partner perm)
        ((3, 4, 5), (False, True, True), (0, 1, 2), (0, 2, 1), None, None),
        ((2, 6, 3, 4, 5), (False, True, False, True, True), (0, 0, 0, 1, 1), (0, 0, 0, 1, 0), None, None),
        ((4, 5, 6), (True, True, True), (0, 2, 2), (0, 2, 2), (4, 6, 5), (0, 2, 1)),
        ((3, 2, 4, 4), (False, True, True, True), (0, 0, 1, 1), (0, 0, 1, 1), (2, 4, 3, 4), (1, 2, 0, 3)),
        ((7,), (True,), (3,), (4,), None, None),
    ]
    for shape, mapped, lo, hi, pshape, perm in cases:
        x = _field(shape, 91).astype(dtype)
        partner = None if pshape is None else _field(pshape, 92).astype(dtype)
        out_shape = [n + (l + h if m else 0) for n, m, l, h in zip(shape, mapped, lo, hi)]
        p_out = int(np.prod([n for n, m in zip(out_shape, mapped) if m]))
        p_in = int(np.prod([n for n, m in zip(shape, mapped) if m]))
        p_partner = 0
        if partner is not None:
            p_partner = int(np.prod([pshape[k] for k in range(len(pshape)) if mapped[perm[k]]]))
        fills = [0.0, -7.25, float("nan")]
        tok = rng.integers(1, p_in + p_partner + 1, size=p_out).astype(np.int64)
        isf = rng.random(p_out) < 0.2
        tok = np.where(isf, T.FILL_BASE + rng.integers(0, len(fills), size=p_out), tok)
        tok = np.where(rng.random(p_out) < 0.3, -tok, tok)
        want = T.gather_tokens(x, partner, tok, mapped, lo, out_shape, fills, perm)
        got = dev.tohost(dev.gather(x, partner, tok, mapped, lo, out_shape, fills, perm))
        assert got.dtype == dtype
        _eq(got, want)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("tpw", [1, 2, 4])
def test_gather_long_rows_tiles_per_wave(dev, dtype, tpw):
    """the row-wise gather kernel (output rows of >= 64 cells; the small cases above go through the per-cell kernel): random
    token maps on the halo frame, interior straight from the input, a partner component, mapped dims with an unmapped one
    between them, odd row lengths, 1 / 2 / 4 tiles per wave-task"""
    from oracle import topology as T
    from xgcm_amd import _hip

    keep = _hip.get_tunable("pad_tpw")
    _hip.set_tunable("pad_tpw", tpw)
    rng = np.random.default_rng(17)
    cases = [
        ((3, 9, 200), (False, True, True), (0, 1, 2), (0, 2, 1), None, None),
        ((2, 5, 131), (False, True, True), (0, 2, 0), (0, 1, 0), None, None),      # innermost dim not padded
        ((2, 6, 3, 70), (True, False, False, True), (1, 0, 0, 3), (1, 0, 0, 2), None, None),
        ((4, 7, 300), (True, True, True), (0, 2, 2), (0, 2, 2), (4, 300, 7), (0, 2, 1)),
        ((1, 3, 515), (False, False, True), (0, 0, 1), (0, 0, 1), None, None),
    ]
    try:
        for shape, mapped, lo, hi, pshape, perm in cases:
            x = _field(shape, 93).astype(dtype)
            partner = None if pshape is None else _field(pshape, 94).astype(dtype)
            out_shape = [n + (l + h if m else 0) for n, m, l, h in zip(shape, mapped, lo, hi)]
            p_out = int(np.prod([n for n, m in zip(out_shape, mapped) if m]))
            p_in = int(np.prod([n for n, m in zip(shape, mapped) if m]))
            p_partner = 0
            if partner is not None:
                p_partner = int(np.prod([pshape[k] for k in range(len(pshape)) if mapped[perm[k]]]))
            fills = [0.0, -7.25, float("nan")]
            tok = rng.integers(1, p_in + p_partner + 1, size=p_out).astype(np.int64)
            tok = np.where(rng.random(p_out) < 0.2, T.FILL_BASE + rng.integers(0, len(fills), size=p_out), tok)
            tok = np.where(rng.random(p_out) < 0.3, -tok, tok)
            want = T.gather_tokens(x, partner, tok, mapped, lo, out_shape, fills, perm)
            _eq(dev.tohost(dev.gather(x, partner, tok, mapped, lo, out_shape, fills, perm)), want)
    finally:
        _hip.set_tunable("pad_tpw", keep)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES + [(2, 40, 2600), (3, 2, 4, 8)])
def test_stencil_with_pregathered_halo(dev, shape, dtype):
    """xg_stencil1d_halo: every kernel family (contiguous V/general, short segments, column chunks)
    must read the low / high halo cell from the halo slab exactly where numpy's concatenate puts it."""
    a = _field(shape, 77).astype(dtype)
    for axis in range(len(shape)):
        for lo, hi in PADS:
            hshape = list(shape)
            hshape[axis] = lo + hi
            halo = _field(hshape, 78).astype(dtype) if lo + hi else np.zeros(hshape, dtype)
            lo_part = np.take(halo, range(0, lo), axis=axis)
            hi_part = np.take(halo, range(lo, lo + hi), axis=axis)
            padded = np.concatenate([lo_part, a, hi_part], axis=axis)
            if padded.shape[axis] < 2:
                continue
            for op in ("diff", "interp"):
                want = R.stencil1d(op, padded, axis, 0, 0, None)
                got = dev.tohost(dev.stencil1d_halo(op, a, halo, axis, lo, hi))
                assert got.dtype == dtype
                _eq(got, want)
            # output metric (derivative): divides after the op, halo or not
            m = (R.synthetic_metric(want.shape, 79)).astype(dtype)
            _eq(dev.tohost(dev.stencil1d_halo("diff", a, halo, axis, lo, hi, m)), R.stencil1d("diff", padded, axis, 0, 0, None) / m)


def test_binary_broadcast(dev):
    a = _field((3, 4, 6, 10), 19)
    for op in ("mul", "div", "add", "sub"):
        for bshape in [(3, 4, 6, 10), (1, 1, 6, 10), (1, 4, 1, 1), (3, 1, 1, 10), (1, 1, 1, 1), (1, 4, 6, 1)]:
            b = R.synthetic_metric(bshape, 23)
            _eq(dev.tohost(dev.binary(op, a, b)), R.binary(op, a, b))
            _eq(dev.tohost(dev.binary(op, b, a)), R.binary(op, b, a))
    a = _field((5, 7), 2)
    b = _field((5, 7), 3)
    _eq(dev.tohost(dev.binary("sub", a, b)), a - b)


@pytest.mark.parametrize("shape", [(3, 9, 64), (2, 70, 33), (5, 6), (2, 2, 130, 258)])
def test_vorticity_fused_equals_unfused_chain(dev, shape):
    u = _field(shape, 51)
    v = _field(shape, 52)
    area2d = R.synthetic_metric((1,) * (len(shape) - 2) + shape[-2:], 53)
    for bc_x, bc_y in itertools.product(BCS, BCS):
        exp = R.vorticity(u, v, area2d, bc_x, bc_y, 0.25, -0.5)
        got = dev.tohost(dev.vorticity(u, v, area2d, bc_x, bc_y, 0.25, -0.5))
        _eq(got, exp)
    # unfused chain through the individual kernels gives the same bits
    dv = dev.stencil1d("diff", v, len(shape) - 1, 1, 0, "fill", 0.0)
    du = dev.stencil1d("diff", u, len(shape) - 2, 1, 0, "fill", 0.0)
    chain = dev.binary("div", dev.binary("sub", dv, du), area2d)
    _eq(dev.tohost(chain), R.vorticity(u, v, area2d, "fill", "fill"))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(3, 9, 64), (2, 70, 33), (5, 6), (2, 2, 130, 258), (4, 1, 2), (3, 5, 4), (1, 1), (2, 3, 1)])
def test_gradient_and_flux_fused_equal_unfused_chains(dev, shape, dtype):
    a = _field(shape, 60).astype(dtype)
    u = _field(shape, 61).astype(dtype)
    v = _field(shape, 62).astype(dtype)
    mx = R.synthetic_metric((1,) * (len(shape) - 2) + shape[-2:], 63).astype(dtype)
    my = R.synthetic_metric(shape, 64).astype(dtype)
    for bc_x, bc_y in itertools.product(("periodic", "fill", "extend"), repeat=2):
        for m1, m2 in ((None, None), (mx, my), (my, None), (None, mx)):
            ex, ey = R.gradient(a, bc_x, bc_y, dtype(0.25), dtype(-0.5), m1, m2)
            gx, gy = dev.gradient(a, bc_x, bc_y, 0.25, -0.5, m1, m2)
            _eq(dev.tohost(gx), ex)
            _eq(dev.tohost(gy), ey)
        ex, ey = R.flux(u, v, a, bc_x, bc_y, dtype(0.25), dtype(-0.5))
        fx, fy = dev.flux(u, v, a, bc_x, bc_y, 0.25, -0.5)
        _eq(dev.tohost(fx), ex)
        _eq(dev.tohost(fy), ey)
    # the chain through the product's own 1-D kernels gives the same bits
    gx, gy = dev.gradient(a, "periodic", "extend")
    _eq(dev.tohost(gx), dev.tohost(dev.stencil1d("diff", a, len(shape) - 1, 1, 0, "periodic")))
    _eq(dev.tohost(gy), dev.tohost(dev.stencil1d("diff", a, len(shape) - 2, 1, 0, "extend")))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(3, 9, 64), (2, 70, 33), (5, 6), (2, 2, 130, 258), (3, 5, 4)])
def test_gradient_and_flux_with_pregathered_halos(dev, shape, dtype):
    """xg_gradient_halo / xg_flux_halo: the column left of i = 0 and the row below j = 0 come from halo
    slabs (what complex topologies gather); either axis may keep an ordinary mode."""
    a = _field(shape, 80).astype(dtype)
    u = _field(shape, 81).astype(dtype)
    v = _field(shape, 82).astype(dtype)
    hx = _field(shape[:-1], 83).astype(dtype)              # (..., Y)
    hy = _field(shape[:-2] + shape[-1:], 84).astype(dtype)  # (..., X)
    m = R.synthetic_metric(shape, 85).astype(dtype)
    ax = np.concatenate([hx[..., None], a], axis=-1)
    ay = np.concatenate([np.expand_dims(hy, -2), a], axis=-2)
    nd = len(shape)
    for bc_x, bc_y in (("halo", "halo"), ("halo", "extend"), ("periodic", "halo"), ("fill", "halo")):
        ex = R.stencil1d("diff", ax, nd - 1, 0, 0, None) if bc_x == "halo" else R.stencil1d("diff", a, nd - 1, 1, 0, bc_x, dtype(0.25))
        ey = R.stencil1d("diff", ay, nd - 2, 0, 0, None) if bc_y == "halo" else R.stencil1d("diff", a, nd - 2, 1, 0, bc_y, dtype(-0.5))
        gx, gy = dev.gradient(a, bc_x, bc_y, 0.25, -0.5, None, None, hx if bc_x == "halo" else None, hy if bc_y == "halo" else None)
        _eq(dev.tohost(gx), ex)
        _eq(dev.tohost(gy), ey)
        gx, gy = dev.gradient(a, bc_x, bc_y, 0.25, -0.5, m, m, hx if bc_x == "halo" else None, hy if bc_y == "halo" else None)
        _eq(dev.tohost(gx), ex / m)
        _eq(dev.tohost(gy), ey / m)
        ix = R.stencil1d("interp", ax, nd - 1, 0, 0, None) if bc_x == "halo" else R.stencil1d("interp", a, nd - 1, 1, 0, bc_x, dtype(0.25))
        iy = R.stencil1d("interp", ay, nd - 2, 0, 0, None) if bc_y == "halo" else R.stencil1d("interp", a, nd - 2, 1, 0, bc_y, dtype(-0.5))
        fx, fy = dev.flux(u, v, a, bc_x, bc_y, 0.25, -0.5, hx if bc_x == "halo" else None, hy if bc_y == "halo" else None)
        _eq(dev.tohost(fx), u * ix)
        _eq(dev.tohost(fy), v * iy)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(3, 7, 9), (2, 5, 131), (5, 3, 2, 67), (1, 4, 1025), (4, 9), (6, 33), (2, 3, 4, 5, 7)])
def test_strided_axis_with_misaligned_rows(dev, shape, dtype):
    """K2g: odd inner extents (rows of the strided axis not 16-B aligned, e.g. `outer` X positions): every op,
    every pad / boundary mode, every strided axis, with and without a pre-gathered halo -- bit for bit."""
    a = _field(shape, 91, nan=True).astype(dtype)
    nd = len(shape)
    for axis in range(nd - 1):
        for op in ("diff", "interp", "min", "max"):
            for lo, hi in ((1, 0), (0, 1), (1, 1), (0, 0)):
                if shape[axis] + lo + hi - 1 < 1:
                    continue
                for bc in (("periodic", "fill", "extend") if lo + hi else (None,)):
                    _eq(dev.tohost(dev.stencil1d(op, a, axis, lo, hi, bc, -1.5)), R.stencil1d(op, a, axis, lo, hi, bc, dtype(-1.5)))
                if lo + hi:
                    hshape = list(shape)
                    hshape[axis] = lo + hi
                    halo = _field(hshape, 92).astype(dtype)
                    padded = np.concatenate([np.take(halo, range(0, lo), axis=axis), a, np.take(halo, range(lo, lo + hi), axis=axis)], axis=axis)
                    _eq(dev.tohost(dev.stencil1d_halo(op, a, halo, axis, lo, hi)), R.stencil1d(op, padded, axis, 0, 0, None))
        # metrics ride along (input metric at the source rows, output metric at the output row), any broadcast pattern
        oshape = list(shape)
        for keep in (set(range(nd)), {axis, nd - 1}, {nd - 1}, {axis}):
            m_in = _metric_for(shape, keep, 95).astype(dtype)
            for (lo, hi), bc in (((1, 0), "periodic"), ((0, 1), "fill"), ((1, 1), "extend")):
                oshape[axis] = shape[axis] + lo + hi - 1
                m_out = _metric_for(tuple(oshape), keep, 96).astype(dtype)
                for mi_, mo_ in ((m_in, m_out), (None, m_out), (m_in, None)):
                    _eq(dev.tohost(dev.stencil1d("interp", a, axis, lo, hi, bc, 0.25, mi_, mo_)),
                        R.stencil1d("interp", a, axis, lo, hi, bc, dtype(0.25), mi_, mo_))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_strided_axis_misaligned_whole_plane_rows(dev, dtype):
    """K2g in its column-chunk order: rows of the strided axis are whole planes with an odd number of cells
    (601 x 901 = 541 501 > 2048 tiles), incl. the groups that straddle two levels and the partial last group."""
    shape = (5, 601, 901)
    a = _field(shape, 93).astype(dtype)
    for op, (lo, hi), bc in (("diff", (1, 0), "fill"), ("interp", (0, 1), "periodic"), ("max", (1, 1), "extend"), ("diff", (0, 0), None)):
        _eq(dev.tohost(dev.stencil1d(op, a, 0, lo, hi, bc, 0.5)), R.stencil1d(op, a, 0, lo, hi, bc, dtype(0.5)))
    halo = _field((1,) + shape[1:], 94).astype(dtype)
    _eq(dev.tohost(dev.stencil1d_halo("diff", a, halo, 0, 1, 0)), R.stencil1d("diff", np.concatenate([halo, a]), 0, 0, 0, None))


def test_gradient_metric_broadcast_patterns(dev):
    shape = (3, 2, 6, 8)
    a = _field(shape, 70)
    for mshape in ((1, 1, 6, 8), (3, 1, 6, 8), (1, 2, 6, 8), (3, 2, 1, 8), (3, 2, 6, 1), (1, 1, 1, 1)):
        m = R.synthetic_metric(mshape, 71)
        ex, ey = R.gradient(a, "fill", "periodic", 0.5, 0.0, m, m)
        gx, gy = dev.gradient(a, "fill", "periodic", 0.5, 0.0, m, m)
        _eq(dev.tohost(gx), ex)
        _eq(dev.tohost(gy), ey)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(3, 9, 64), (2, 70, 33), (5, 6), (2, 2, 130, 258), (4, 1, 2), (3, 5, 4)])
def test_divergence_fused_equals_unfused_chain(dev, shape, dtype):
    u = _field(shape, 61).astype(dtype)
    v = _field(shape, 62).astype(dtype)
    area2d = R.synthetic_metric((1,) * (len(shape) - 2) + shape[-2:], 63).astype(dtype)
    for bc_x, bc_y in itertools.product(BCS, BCS):
        exp = R.divergence(u, v, area2d, bc_x, bc_y, dtype(0.25), dtype(-0.5))
        got = dev.tohost(dev.divergence(u, v, area2d, bc_x, bc_y, 0.25, -0.5))
        assert got.dtype == dtype
        _eq(got, exp)
        _eq(dev.tohost(dev.divergence(u, v, None, bc_x, bc_y, 0.25, -0.5)),
            R.divergence(u, v, np.ones((1,) * len(shape), dtype=dtype), bc_x, bc_y, dtype(0.25), dtype(-0.5)))
    du = dev.stencil1d("diff", u, len(shape) - 1, 0, 1, "periodic", 0.0)
    dv = dev.stencil1d("diff", v, len(shape) - 2, 0, 1, "extend", 0.0)
    chain = dev.binary("div", dev.binary("add", du, dv), area2d)
    _eq(dev.tohost(chain), dev.tohost(dev.divergence(u, v, area2d, "periodic", "extend")))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_vorticity_divergence_area_broadcast_patterns(dev, dtype):
    """area with only SOME of the leading dims (rAz(face, j, i) for a (time, Z, face, j, i) field, ...)."""
    shape = (2, 3, 4, 6, 64)
    u = _field(shape, 71).astype(dtype)
    v = _field(shape, 72).astype(dtype)
    for ashape in [(1, 1, 4, 6, 64), (1, 3, 1, 6, 64), (2, 1, 4, 6, 64), (2, 3, 4, 6, 64), (1, 1, 1, 6, 64), (1, 3, 4, 6, 64),
                   (2, 1, 1, 6, 64), (1, 1, 4, 1, 64), (1, 1, 4, 6, 1)]:
        area = R.synthetic_metric(ashape, 73).astype(dtype)
        _eq(dev.tohost(dev.vorticity(u, v, area, "periodic", "extend")), R.vorticity(u, v, area, "periodic", "extend"))
        _eq(dev.tohost(dev.divergence(u, v, area, "fill", "periodic", 0.5, 0.0)), R.divergence(u, v, area, "fill", "periodic", dtype(0.5), dtype(0.0)))


def test_synthetic_bit_identical(dev):
    for n, seed, off in [(1000, 1, 0), (4097, 4, 123456789), (10, 53, 2**40)]:
        got = dev.tohost(dev.synthetic((n,), seed, off))
        _eq(got, R.synthetic(n, seed, off))
    got = dev.tohost(dev.synthetic((33, 5), 31, 0, 1000.0, 1000.0))
    _eq(got, R.synthetic_metric((33, 5), 31))


def test_c_abi_rejects_bad_arguments(dev):
    from xgcm_amd import _hip

    a = dev.asdevice(_field((4, 8), 1))
    with pytest.raises(_hip.XgcmHipError, match="no boundary mode"):
        dev.stencil1d("diff", a, 1, 1, 0, None)
    # an axis trimmed away completely: constant halo is fine, wrap/extend of nothing is an error (numpy.pad)
    one = _field((4, 1), 1)
    _eq(dev.tohost(dev.cumsum1d(one, 1, 0, 1, 1, 0, "fill", 2.5)), R.cumsum1d(one, 1, 0, 1, 1, 0, "fill", 2.5))
    assert dev.tohost(dev.cumsum1d(one, 1, 1, 0, 0, 0, "fill")).shape == (4, 0)
    with pytest.raises(ValueError, match="can't extend empty axis"):
        dev.cumsum1d(one, 1, 0, 1, 1, 0, "extend")
    with pytest.raises(ValueError, match="can't extend empty axis"):
        R.cumsum1d(one, 1, 0, 1, 1, 0, "extend")


def test_library_device_helpers_roundtrip():
    """xg_malloc / h2d / kernel / d2h / events without torch: the path a non-torch host would use."""
    import ctypes as C

    from xgcm_amd import _hip

    lib = _hip.load()
    assert lib.xg_device_count() >= 1
    a = R.synthetic_field((16, 128), 2)
    nbytes = a.nbytes
    din, dout = C.c_void_p(), C.c_void_p()
    _hip.check(lib.xg_malloc(C.byref(din), nbytes))
    _hip.check(lib.xg_malloc(C.byref(dout), nbytes))
    e0, e1 = C.c_void_p(), C.c_void_p()
    _hip.check(lib.xg_event_create(C.byref(e0)))
    _hip.check(lib.xg_event_create(C.byref(e1)))
    _hip.check(lib.xg_memcpy_h2d(din, a.ctypes.data, nbytes, None))
    _hip.check(lib.xg_event_record(e0, None))
    _hip.check(lib.xg_stencil1d_f64(0, din, dout, _hip.i64(a.shape), 2, 1, 128, 1, 0, 1, 0.0, None, None, None, None, None))
    _hip.check(lib.xg_event_record(e1, None))
    out = np.empty_like(a)
    _hip.check(lib.xg_memcpy_d2h(out.ctypes.data, dout, nbytes, None))
    _hip.check(lib.xg_stream_sync(None))
    ms = C.c_float()
    _hip.check(lib.xg_event_elapsed_ms(e0, e1, C.byref(ms)))
    assert ms.value >= 0
    _eq(out, R.stencil1d("diff", a, 1, 1, 0, "periodic"))
    for h in (din, dout):
        _hip.check(lib.xg_free(h))
    for e in (e0, e1):
        _hip.check(lib.xg_event_destroy(e))


def test_degenerate_shapes(dev):
    """empty outer dims, length-1 axes, 0-d results: the edge cases of the shape logic."""
    import torch

    # length-1 axis: center->left periodic wraps onto itself, extend replicates, fill uses the constant
    a = _field((3, 1, 4), 1)
    _eq(dev.tohost(dev.stencil1d("diff", a, 1, 1, 0, "periodic")), np.zeros((3, 1, 4)))
    _eq(dev.tohost(dev.stencil1d("interp", a, 1, 0, 1, "extend")), a)
    _eq(dev.tohost(dev.stencil1d("diff", a, 1, 1, 0, "fill", 2.0)), a - 2.0)
    _eq(dev.tohost(dev.stencil1d("diff", a, 1, 1, 1, "fill", 2.0)), R.stencil1d("diff", a, 1, 1, 1, "fill", 2.0))
    # empty leading dim: nothing to do, shape bookkeeping still right
    e = torch.empty((0, 5, 8), dtype=torch.float64, device="cuda")
    assert tuple(dev.stencil1d("diff", e, 2, 1, 0, "periodic").shape) == (0, 5, 8)
    assert tuple(dev.stencil1d("diff", e, 1, 1, 1, "fill").shape) == (0, 6, 8)
    assert tuple(dev.cumsum1d(e, 1, 0, 0, 1, 0, "fill").shape) == (0, 6, 8)
    assert tuple(dev.reduce1d(e, 1).shape) == (0, 8)
    # reducing a 1-D array gives a 0-d result
    v = _field((1000,), 3)
    got = dev.tohost(dev.reduce1d(v, 0))
    assert got.shape == () and abs(got - v.sum()) < 1e-12
    # center->inner on a length-2 axis leaves one cell
    b = _field((2, 6), 4)
    _eq(dev.tohost(dev.stencil1d("diff", b, 0, 0, 0, None)), b[1:] - b[:-1])


@pytest.mark.parametrize("shape", [(3, 9, 64), (2, 70, 34), (6, 8), (2, 2, 5, 258), (1, 4, 2)])
@pytest.mark.parametrize("op", OPS)
def test_stencil2d_equals_two_sequential_passes(dev, shape, op):
    """one fused launch == reference order of operations: op along the first axis, then the second."""
    a = _field(shape, 61, nan=op in ("min", "max"))
    ax_x, ax_y = len(shape) - 1, len(shape) - 2
    for order in (0, 1):
        for padx, pady in itertools.product([(1, 0), (0, 1)], [(1, 0), (0, 1)]):
            for bc_x, bc_y in itertools.product(BCS, BCS):
                if order == 0:
                    t = R.stencil1d(op, a, ax_x, *padx, bc_x, 0.75)
                    exp = R.stencil1d(op, t, ax_y, *pady, bc_y, -1.5)
                else:
                    t = R.stencil1d(op, a, ax_y, *pady, bc_y, -1.5)
                    exp = R.stencil1d(op, t, ax_x, *padx, bc_x, 0.75)
                assert dev.stencil2d_supported(dev.asdevice(a), padx, pady)
                got = dev.tohost(dev.stencil2d(op, a, order, padx, bc_x, 0.75, pady, bc_y, -1.5))
                _eq(got, exp)
    assert not dev.stencil2d_supported(dev.asdevice(_field((3, 5, 33), 1)), (1, 0), (1, 0))  # odd nx
    assert not dev.stencil2d_supported(dev.asdevice(a), (1, 1), (1, 0))  # length-changing pair


def test_c_abi_from_plain_c(tmp_path):
    """examples/c_abi_demo.c: the boundary is a real C ABI -- compile with gcc, link the .so, run."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "c_abi_demo.c"), "-I" + os.path.join(root, "include"),
                           "-L" + os.path.join(root, "xgcm_amd"), "-lxgcm_hip", "-Wl,-rpath," + os.path.join(root, "xgcm_amd"),
                           "-lm", "-o", exe])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 mismatching cells" in res.stdout


def test_binary_with_more_dims_than_the_abi_addresses(dev):
    """`interp(a, X) * derivative(b, Y)` of fields on different positions of three axes (+ a record dim) is a 9-D outer
    product by name: more dims than xg_binary takes (8).  Found by tools/gpu_grid_fuzz_sweep.py; numpy is the expectation."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal((2, 3, 1, 2, 1, 3, 1, 2, 1))
    b = rng.standard_normal((1, 1, 2, 1, 3, 1, 2, 1, 2))
    for op, f in (("mul", np.multiply), ("add", np.add), ("sub", np.subtract), ("div", np.divide)):
        _eq(dev.tohost(dev.binary(op, a, b)), f(a, b))
    c = rng.standard_normal((2, 1, 1, 1, 1, 1, 1, 1, 3))  # nine dims, two real ones: merged, one launch
    _eq(dev.tohost(dev.binary("mul", c, c)), c * c)
