"""A chained scan / reduction whose hand-off gives up must never be consumable as a wrong result (VERDICT r02 weak 1,
reference contract xgcm/grid.py:1316: the cumulative sum is simply right).

`scan_chain_spin = 1` makes nearly every chunk give up after ONE poll of its predecessor's slot.  What must hold:
the array read back equals the oracle (the marching twin queued behind every chained launch redid the call in stream),
the event is reported where the result is read (a ChainRescueWarning from tohost / .values, xg_chain_status), the
library plans marches while the sticky word is set, and after xg_chain_rearm() the chained kernels run again on a
clean workspace (the rescue scrubbed the stale epochs a late predecessor left behind)."""
import warnings

import numpy as np
import pytest

from oracle import refimpl as R

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture()
def starved():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xgcm_amd import _hip
    from xgcm_amd import device as D

    keep = {k: _hip.get_tunable(k) for k in ("scan_chain_spin", "scan_chain", "reduce_zl", "reduce_ldsw")}
    _hip.set_tunable("reduce_ldsw", 0)  # these tests are about the CHAINED reduction, not the level-sharing marches (K4L, K4Z) that replaced it by default
    torch.cuda.synchronize()
    _hip.chain_rearm()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _hip.chain_check()  # swallow anything an earlier test left unreported
    yield D, _hip
    torch.cuda.synchronize()
    for k, v in keep.items():
        _hip.set_tunable(k, v)
    _hip.chain_rearm()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _hip.chain_check()


def _field(shape, seed, dtype, nan=True):
    a = R.synthetic_field(shape, seed).astype(dtype)
    if nan:
        a.reshape(-1)[[5, a.size // 3, a.size - 2]] = np.nan
    return a


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_starved_chained_scan_is_redone_and_reported(starved, dtype):
    D, _hip = starved
    shape = (3, 2400, 256)  # 75 chunks of 32 rows per column
    a = _field(shape, 5, dtype)
    m_in = R.synthetic_metric((1,) + shape[1:], 43).astype(dtype)
    cases = [dict(args=(0, 1, 1, 0, "extend", 0.0, False, True), m=(None, None)),
             dict(args=(1, 0, 0, 1, "periodic", 0.0, True, True), m=(m_in, None)),
             dict(args=(0, 0, 1, 0, "fill", 0.5, False, False), m=(None, None))]
    _hip.set_tunable("scan_chain_spin", 1)
    redone_before = _hip.chain_status()[1]
    for k, case in enumerate(cases):
        _hip.chain_rearm()
        tl, th, pl, ph, bc, fill, rev, skip = case["args"]
        want = R.cumsum1d(a, 1, tl, th, pl, ph, bc, dtype(fill), rev, skip, *case["m"])
        out = D.cumsum1d(a, 1, tl, th, pl, ph, bc, fill, rev, skip, *case["m"])
        with pytest.warns(_hip.ChainRescueWarning, match="redone by the marching kernel"):
            got = D.tohost(out)  # the event surfaces here, before the values are handed over
        assert np.array_equal(got, want, equal_nan=True)
        gave_up, redone = _hip.chain_status()
        assert gave_up == 1 and redone == redone_before + k + 1
    # sticky: the library plans marches now -- no further rescue, still the same bits
    tl, th, pl, ph, bc, fill, rev, skip = cases[0]["args"]
    want = R.cumsum1d(a, 1, tl, th, pl, ph, bc, dtype(fill), rev, skip)
    with warnings.catch_warnings():
        warnings.simplefilter("error", _hip.ChainRescueWarning)
        assert np.array_equal(D.tohost(D.cumsum1d(a, 1, tl, th, pl, ph, bc, fill, rev, skip)), want, equal_nan=True)
    assert _hip.chain_status()[1] == redone_before + len(cases)
    # re-armed with the normal spin: the chained kernel again, on a workspace the rescue left all-zero
    _hip.set_tunable("scan_chain_spin", 1 << 22)
    _hip.chain_rearm()
    with warnings.catch_warnings():
        warnings.simplefilter("error", _hip.ChainRescueWarning)
        for _ in range(3):
            assert np.array_equal(D.tohost(D.cumsum1d(a, 1, tl, th, pl, ph, bc, fill, rev, skip)), want, equal_nan=True)
    assert _hip.chain_status() == (0, redone_before + len(cases))


@pytest.mark.parametrize("zl", [1, 2, 4])
def test_starved_chained_reduction_is_redone_and_reported(starved, zl):
    D, _hip = starved
    shape = (4, 2400, 256)
    a = _field(shape, 6, np.float64)
    w = R.synthetic_metric((1,) + shape[1:], 44)
    _hip.set_tunable("reduce_zl", zl)
    _hip.set_tunable("scan_chain", 0)
    ref = {mode: D.tohost(D.reduce1d(a, 1, w, mode)) for mode in (True, "mean_valid", "pair_all")}
    _hip.set_tunable("scan_chain", 1)
    _hip.set_tunable("scan_chain_spin", 1)
    for mode, want in ref.items():
        _hip.chain_rearm()
        before = _hip.chain_status()[1]
        out = D.reduce1d(a, 1, w, mode)
        with pytest.warns(_hip.ChainRescueWarning):
            got = D.tohost(out)
        assert np.array_equal(got, want, equal_nan=True)
        assert _hip.chain_status() == (1, before + 1)
    assert np.array_equal(ref[True], R.integrate(a, 1, w, True), equal_nan=True)


def test_starved_chain_inside_a_graph_replay(starved):
    """The rescue kernel is part of the captured graph: replays whose chained scan gives up still deliver the march's
    bits, and the event is reported at the next replay / read."""
    D, _hip = starved
    from xgcm_amd.graphs import capture

    shape = (2, 2048, 128)
    buf = D.asdevice(_field(shape, 9, np.float64, nan=False))
    _hip.set_tunable("scan_chain_spin", 1)

    def scan():
        _hip.chain_rearm()  # the warm-up run gives up and sets the sticky word: the CAPTURED launch must be the chained one
        return D.cumsum1d(buf, 1, 0, 1, 1, 0, "extend", 0.0, False, True)

    step = capture(scan, warmup=1)
    _hip.chain_rearm()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _hip.chain_check()
    before = _hip.chain_status()[1]
    for seed in (3, 4):
        new = _field(shape, seed, np.float64, nan=False)
        buf.copy_(torch.from_numpy(new).cuda())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = step()
            got = D.tohost(out)
        assert np.array_equal(got, R.cumsum1d(new, 1, 0, 1, 1, 0, "extend", 0.0, False, True))
    assert _hip.chain_status()[1] >= before + 1


def test_destroying_a_stream_releases_its_chain_workspace(starved):
    """a destroyed stream's workspace entry goes with it; a later stream (possibly the same handle) starts
    from a fresh, zeroed one and computes the same bits."""
    import ctypes

    D, _hip = starved
    lib = _hip.load()
    a = D.asdevice(_field((2, 640, 128), 12, np.float64))
    want = R.cumsum1d(D.tohost(a), 1, 0, 0, 0, 0, None, 0.0, False, True)
    for _ in range(4):
        h = ctypes.c_void_p()
        _hip.check(lib.xg_stream_create(ctypes.byref(h)))
        s = torch.cuda.ExternalStream(h.value)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out = D.cumsum1d(a, 1, 0, 0, 0, 0, None, 0.0, False, True)
        s.synchronize()
        assert np.array_equal(D.tohost(out), want, equal_nan=True)
        _hip.check(lib.xg_stream_destroy(h))
    assert lib.xg_stream_destroy(None) != 0  # the null stream is not the caller's to destroy
