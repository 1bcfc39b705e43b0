#!/usr/bin/env python3
"""What the host <-> HBM link gives on this box, next to what xgcm_amd.streaming gets out of it.

One JSON line per measurement: hipHostMalloc'ed (torch pin_memory) buffers H2D / D2H alone and both at once, the same from
numpy arrays page-locked in place (hipHostRegister: its own cost reported separately, touched and untouched pages), and
`stream_records` on the same bytes split into its phases."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def rate(nbytes, secs):
    return round(nbytes / secs / 1e9, 2)


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    n = int(gb * 1e9 / 8)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    d_in = torch.empty(n, dtype=torch.float64, device=dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    t0 = time.perf_counter()
    h_in = torch.empty(n, dtype=torch.float64).pin_memory()
    h_out = torch.empty(n, dtype=torch.float64).pin_memory()
    print(json.dumps({"what": "hipHostMalloc (torch pin_memory) of 2 buffers", "GB_each": gb, "s": round(time.perf_counter() - t0, 3)}), flush=True)
    h_in.fill_(1.5)

    def h2d():
        with torch.cuda.stream(s1):
            d_in.copy_(h_in, non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            h_out.copy_(d_out, non_blocking=True)

    def both():
        h2d()
        d2h()

    nb = n * 8
    print(json.dumps({"what": "pinned H2D alone", "GBps": rate(nb, timed(h2d))}), flush=True)
    print(json.dumps({"what": "pinned D2H alone", "GBps": rate(nb, timed(d2h))}), flush=True)
    print(json.dumps({"what": "pinned H2D + D2H at once (each way)", "GBps_each_way": rate(nb, timed(both))}), flush=True)
    # in blocks of 1/8: what a pipeline of 8 records sees
    k = n // 8

    def both_blocks():
        for i in range(8):
            with torch.cuda.stream(s1):
                d_in[i * k:(i + 1) * k].copy_(h_in[i * k:(i + 1) * k], non_blocking=True)
            with torch.cuda.stream(s2):
                h_out[i * k:(i + 1) * k].copy_(d_out[i * k:(i + 1) * k], non_blocking=True)

    print(json.dumps({"what": "the same in 8 blocks per direction", "GBps_each_way": rate(nb, timed(both_blocks))}), flush=True)
    del h_in, h_out

    rt = torch.cuda.cudart()
    for touched in (False, True):
        a = np.empty(n, dtype=np.float64)
        if touched:
            a[:] = 2.5
        t = torch.from_numpy(a)
        t0 = time.perf_counter()
        err = rt.cudaHostRegister(t.data_ptr(), n * 8, 0)
        dt_reg = time.perf_counter() - t0
        print(json.dumps({"what": f"hipHostRegister of a numpy array, pages {'touched' if touched else 'never touched'}", "rc": int(err),
                          "s": round(dt_reg, 3), "GBps": rate(nb, dt_reg)}), flush=True)

        def h2d_reg():
            with torch.cuda.stream(s1):
                d_in.copy_(t, non_blocking=True)

        def d2h_reg():
            with torch.cuda.stream(s2):
                t.copy_(d_out, non_blocking=True)

        print(json.dumps({"what": "registered numpy H2D alone", "GBps": rate(nb, timed(h2d_reg))}), flush=True)
        print(json.dumps({"what": "registered numpy D2H alone", "GBps": rate(nb, timed(d2h_reg))}), flush=True)
        t0 = time.perf_counter()
        rt.cudaHostUnregister(t.data_ptr())
        print(json.dumps({"what": "hipHostUnregister", "s": round(time.perf_counter() - t0, 3)}), flush=True)
        del t, a
    # pageable copies for reference
    a = np.full(n, 3.5)
    t = torch.from_numpy(a)
    print(json.dumps({"what": "pageable numpy H2D (torch .to)", "GBps": rate(nb, timed(lambda: d_in.copy_(t)))}), flush=True)
    print(json.dumps({"what": "pageable numpy D2H", "GBps": rate(nb, timed(lambda: t.copy_(d_out)))}), flush=True)
    del d_in, d_out, a, t

    # the product's pipeline on 8 records, phases separated
    from xgcm_amd import device as D
    from xgcm_amd.streaming import stream_blocks, stream_records
    import threading

    def touch(a, lo, hi):
        a[lo:hi:512] = 0.0

    for T in (1, 4, 8):
        a = np.empty(int(4e9 // 8))
        t0 = time.perf_counter()
        cuts = [a.size * i // T for i in range(T + 1)]
        th = [threading.Thread(target=touch, args=(a, cuts[i], cuts[i + 1])) for i in range(T)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        print(json.dumps({"what": f"first touch of 4 GB from numpy.empty by {T} thread(s), one write per page", "s": round(dt, 3), "GBps": rate(4e9, dt)}), flush=True)
        del a

    nz, ny, nx = 25, 2400, 3600
    src = np.random.default_rng(0).standard_normal((8, nz, ny, nx))
    fn = lambda x: D.stencil1d("diff", x, 3, 1, 0, "periodic")  # noqa: E731
    nbs = src.nbytes
    for label, reg, fresh_out in (("first call, fresh output array (untouched pages)", True, True), ("second call, same arrays", True, False),
                                  ("third call", True, False), ("register=False: pageable copies, D2H in a worker thread", False, False)):
        if fresh_out:
            out = np.empty_like(src)
        t0 = time.perf_counter()
        stream_records(fn, src, block=1, out=out, register=reg)
        dt = time.perf_counter() - t0
        print(json.dumps({"what": f"stream_records, 8 records of {nbs / 8e9:.2f} GB: {label}", "s": round(dt, 3), "GBps_each_way": rate(nbs, dt)}), flush=True)
        assert np.array_equal(out[-1], src[-1] - np.roll(src[-1], 1, axis=-1))
    # numpy in -> numpy out through the device layer (what Grid.diff does with a large host array): block size of the cut
    big = src.reshape(8 * nz, ny, nx)[:75]
    big = np.ascontiguousarray(big)
    default = D.HOST_STREAM_BLOCK_BYTES
    kept = [D.stencil1d("diff", big, 2, 1, 0, "periodic")]
    for mb in (None, 64, 256, 1024):
        D.HOST_STREAM_BLOCK_BYTES = default if mb is None else mb << 20
        t0 = time.perf_counter()
        res = D.stencil1d("diff", big, 2, 1, 0, "periodic")
        dt = time.perf_counter() - t0
        kept.append(res)  # (released outside the timing)
        label = "the default (a sixth of the array)" if mb is None else f"{mb} MB"
        print(json.dumps({"what": f"device.stencil1d on a 5.2 GB host array, numpy in -> numpy out, blocks of {label}", "s": round(dt, 3),
                          "GBps_each_way": rate(big.nbytes, dt)}), flush=True)
    D.HOST_STREAM_BLOCK_BYTES = default
    del kept
    assert np.array_equal(res[-1], big[-1] - np.roll(big[-1], 1, axis=-1))
    del res
    ro = src.view()
    ro.flags.writeable = False
    t0 = time.perf_counter()
    last = []
    stream_blocks(fn, (ro[i:i + 1] for i in range(8)), sink=lambda k, r: last.append(r[0, 0, 0, :8].copy()))
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "stream_blocks: the same records as an iterable of host blocks (page-locked in place block by block)", "s": round(dt, 3),
                      "GBps_each_way": rate(nbs, dt)}), flush=True)
    assert np.array_equal(last[-1], (src[-1] - np.roll(src[-1], 1, axis=-1))[0, 0, :8])

if __name__ == "__main__":
    main()
