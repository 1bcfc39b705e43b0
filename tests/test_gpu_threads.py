"""The threading clause of the boundary: "ABI re-entrant & thread-safe; explicit `stream` argument; no
global mutable state besides the lazily-created per-device context".  It is the reference's execution model under dask:
`dask="parallelized"` invokes the ufunc concurrently from scheduler threads on disjoint blocks (xgcm/grid.py:786-789,
xgcm/grid_ufunc.py:966-984).

Host threads -- each on its own `xg_stream_create` stream, and in a second variant all on the SAME stream -- run plain
and metric stencils, chained scans (`k_cumsum_chain`, whose hand-off workspace is per (device, stream) under a mutex),
marching scans, weighted reductions, fused vorticity and a deliberately failing call, many times over.  Every result
must equal the single-threaded result bit for bit, every thread must read back ITS OWN error text from
`xg_last_error` (thread-local buffer), and `xg_chain_status` must stay clean (no hand-off gave up, nothing was redone).
ctypes releases the GIL for the duration of every ABI call, so the calls genuinely overlap.

`xg_set_tunable` is the one documented exception (include/xgcm_hip.h): process-wide launch-shape switches for A/B
measurements, to be set while no other thread is inside the library; the test asserts they are not touched here.
"""

import ctypes
import threading

import numpy as np
import pytest

from oracle import refimpl as R

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N_THREADS = 6
ROUNDS = 12


def _work(D, inputs, tid):
    """the operator mix of one thread; returns host copies"""
    a, b, m2, mz, area, mb = inputs
    out = {}
    out["diff_x"] = D.stencil1d("diff", a, 2, 1, 0, "periodic")
    out["interp_y_m"] = D.stencil1d("interp", a, 1, 1, 0, "extend", 0.0, m2, m2)
    out["deriv_z"] = D.stencil1d("diff", a, 0, 1, 0, "fill", 0.5, None, mz)
    out["cumsum_y_chain"] = D.cumsum1d(b, 1, 0, 1, 1, 0, "fill", 0.0, False, True)          # columns of 2400 rows: chained
    out["cumint_y_chain"] = D.cumsum1d(b, 1, 0, 0, 0, 0, None, 0.0, True, True, mb, None)   # reversed, weighted: chained
    out["cumsum_z"] = D.cumsum1d(a, 0, 0, 0, 1, 0, "fill", 0.0, False, True)
    out["cumsum_x"] = D.cumsum1d(a, 2, 0, 1, 1, 0, "extend", 0.0, False, True)
    out["integrate_z"] = D.reduce1d(a, 0, mz, True)
    out["average_y"] = D.reduce1d(a, 1, m2, "mean_valid")
    out["vorticity"] = D.vorticity(a, a, area, "fill", "fill", 0.0, 0.0)
    return {k: v.clone() for k, v in out.items()}


def _inputs(D, tid, seed0=900):
    nz, ny, nx = 6, 96, 256
    a = D.synthetic((nz, ny, nx), seed0 + tid)
    b = D.synthetic((2, 2400, 64), seed0 + 50 + tid)           # 75 chunks of 32 rows per column: the chained scan
    m2 = D.synthetic((1, ny, nx), seed0 + 100 + tid, 0, 1000.0, 1000.0)
    mz = D.synthetic((nz, 1, 1), seed0 + 200 + tid, 0, 1000.0, 1000.0)
    area = D.synthetic((1, ny, nx), seed0 + 300 + tid, 0, 1000.0, 1000.0)
    mb = D.synthetic((1, 2400, 64), seed0 + 400 + tid, 0, 1000.0, 1000.0)
    return a, b, m2, mz, area, mb


@pytest.mark.parametrize("own_streams", [True, False], ids=["one stream per thread", "all threads on one stream"])
def test_concurrent_host_threads_get_single_thread_results(own_streams):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xgcm_amd import _hip
    from xgcm_amd import device as D

    lib = _hip.load()
    tun_before = {k: _hip.get_tunable(k) for k in ("scan_chain", "scan_chain_spin", "reduce_ldsw")}
    torch.cuda.synchronize()
    _hip.chain_rearm()
    gave_up0, redone0 = _hip.chain_status()
    inputs = [_inputs(D, t) for t in range(N_THREADS)]
    torch.cuda.synchronize()
    want = [{k: v.cpu().numpy() for k, v in _work(D, inputs[t], t).items()} for t in range(N_THREADS)]  # one thread
    torch.cuda.synchronize()

    handles, streams = [], []
    shared = None
    if own_streams:
        for _ in range(N_THREADS):
            h = ctypes.c_void_p()
            _hip.check(lib.xg_stream_create(ctypes.byref(h)))
            handles.append(h)
            streams.append(torch.cuda.ExternalStream(h.value))
    else:
        h = ctypes.c_void_p()
        _hip.check(lib.xg_stream_create(ctypes.byref(h)))
        handles.append(h)
        shared = torch.cuda.ExternalStream(h.value)
        streams = [shared] * N_THREADS
    dev = torch.cuda.current_device()
    start = threading.Barrier(N_THREADS)
    failures, errors_seen = [], [None] * N_THREADS

    def body(t):
        try:
            torch.cuda.set_device(dev)
            start.wait()
            with torch.cuda.stream(streams[t]):
                for r in range(ROUNDS):
                    got = _work(D, inputs[t], t)
                    # a call that fails, with a text only this thread can have produced (elem_bytes = 100 + t)
                    rc = lib.xg_bswap(ctypes.c_void_p(inputs[t][0].data_ptr()), 16, 100 + t, ctypes.c_void_p(streams[t].cuda_stream))
                    msg = _hip.last_error()
                    if rc == 0 or f"byte swap of {100 + t}-byte elements" not in msg:
                        failures.append((t, r, "error text", rc, msg))
                    errors_seen[t] = msg
                    streams[t].synchronize()
                    for k, v in got.items():
                        if not np.array_equal(v.cpu().numpy(), want[t][k], equal_nan=True):
                            failures.append((t, r, k))
        except Exception as exc:  # noqa: BLE001
            failures.append((t, "exception", repr(exc)))

    threads = [threading.Thread(target=body, args=(t,)) for t in range(N_THREADS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not failures, failures[:5]
    assert all(m is not None and f"{100 + t}-byte" in m for t, m in enumerate(errors_seen))
    gave_up, redone = _hip.chain_status()
    assert (gave_up, redone) == (gave_up0, redone0) == (0, redone0), "a chained hand-off gave up under concurrency"
    assert {k: _hip.get_tunable(k) for k in tun_before} == tun_before
    for h in handles:
        _hip.check(lib.xg_stream_destroy(h))


def test_grid_operators_from_threads_share_one_grid():
    """the labelled layer: one `Grid` (its metric cache, its deferred-fusion switch) used from several threads at once"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from xgcm_amd import DataArray, Dataset, Grid

    nz, ny, nx = 4, 48, 128
    coords = {"XC": ("XC", np.arange(nx) + 0.5), "XG": ("XG", np.arange(nx) * 1.0), "YC": ("YC", np.arange(ny) + 0.5),
              "YG": ("YG", np.arange(ny) * 1.0), "Z": ("Z", np.arange(nz) + 0.5), "Zl": ("Zl", np.arange(nz) * 1.0)}
    ds = Dataset({"dxC": (("YC", "XG"), R.synthetic_metric((ny, nx), 31)), "drF": (("Z",), R.synthetic_metric((nz,), 33)),
                  "rAz": (("YG", "XG"), R.synthetic_metric((ny, nx), 53))}, coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "fill", "Z": "fill"}, metrics={("X",): ["dxC"], ("Z",): ["drF"]},
                autoparse_metadata=False)
    fields = [R.synthetic_field((nz, ny, nx), 700 + t) for t in range(N_THREADS)]

    def ops(t, fuse):
        T = DataArray(fields[t], ("Z", "YC", "XC")).to_device()
        U = DataArray(fields[t], ("Z", "YC", "XG")).to_device()
        V = DataArray(fields[(t + 1) % N_THREADS], ("Z", "YG", "XC")).to_device()
        res = [grid.derivative(T, "X"), grid.cumsum(T, "Z"), grid.integrate(T, "Z"), grid.interp(T, ["X", "Y"])]
        if fuse:
            with grid.fused():
                zeta = (grid.diff(V, "X") - grid.diff(U, "Y")) / ds["rAz"]
                inside = grid._fusing
            res.append(zeta)
            assert inside
        else:
            assert not grid._fusing        # another thread's `with grid.fused()` is not this thread's
            res.append((grid.diff(V, "X") - grid.diff(U, "Y")) / ds["rAz"])
        return [np.asarray(r.values) for r in res]

    want = [ops(t, False) for t in range(N_THREADS)]
    failures = []
    start = threading.Barrier(N_THREADS)
    dev = torch.cuda.current_device()

    def body(t):
        try:
            torch.cuda.set_device(dev)
            start.wait()
            for r in range(6):
                got = ops(t, fuse=(t + r) % 2 == 0)
                for g, w in zip(got, want[t]):
                    if not np.array_equal(g, w, equal_nan=True):
                        failures.append((t, r))
        except Exception as exc:  # noqa: BLE001
            failures.append((t, repr(exc)))

    threads = [threading.Thread(target=body, args=(t,)) for t in range(N_THREADS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not failures, failures[:5]
