#!/usr/bin/env python3
"""Does a kernel's rate depend on WHERE its buffers lie?  (round 4: cumsum Z measured 1.69 ms in one process and 1.90 ms in
another of the same session, same box, same kernel -- tools/survey.py --trace against tools/roofline_table.py.)

Times one operator through the raw C ABI on buffers carved out of ONE large allocation at chosen byte offsets between input
and output, then on fresh allocations (allocator cache emptied in between), and prints one JSON line per placement.

    python tools/addr_probe.py [--op cumZ|cumX|cumY|diffZ|copy] [--shape 75,2400,3600] [--reps 7]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from xgcm_amd import _hip  # noqa: E402
from xgcm_amd import device as D  # noqa: E402


def launch(lib, op, src, dst, shape, stream):
    sh = _hip.i64(shape)
    if op == "cumZ":
        return lib.xg_cumsum1d_f64(src, dst, sh, 3, 0, 0, 1, 0, 1, 1, 0, _hip.BC["fill"], 0.0, None, None, None, None, stream)
    if op == "cumY":
        return lib.xg_cumsum1d_f64(src, dst, sh, 3, 1, 0, 1, 0, 1, 1, 0, _hip.BC["fill"], 0.0, None, None, None, None, stream)
    if op == "cumX":
        return lib.xg_cumsum1d_f64(src, dst, sh, 3, 2, 0, 1, 0, 1, 1, 0, _hip.BC["fill"], 0.0, None, None, None, None, stream)
    if op == "diffZ":
        return lib.xg_stencil1d_f64(0, src, dst, sh, 3, 0, shape[0], 1, 0, _hip.BC["fill"], 0.0, None, None, None, None, stream)
    if op == "diffY":
        return lib.xg_stencil1d_f64(0, src, dst, sh, 3, 1, shape[1], 1, 0, _hip.BC["extend"], 0.0, None, None, None, None, stream)
    raise SystemExit(f"unknown op {op}")


def time_it(lib, op, src, dst, shape, reps):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        _hip.check(launch(lib, op, src, dst, shape, st))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        _hip.check(launch(lib, op, src, dst, shape, st))
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="cumZ")
    ap.add_argument("--shape", default="75,2400,3600")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--fresh", type=int, default=6, help="rounds on fresh allocations")
    a = ap.parse_args()
    shape = [int(v) for v in a.shape.split(",")]
    n = shape[0] * shape[1] * shape[2]
    nbytes = n * 8
    lib = _hip.load()
    alg = 2.0 * nbytes
    offsets = [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 5 << 20, 8 << 20, 17 << 20, 32 << 20, (32 << 20) + 65536, 33 << 20,
               64 << 20, 100 << 20, 128 << 20]
    big = torch.empty(2 * nbytes + max(offsets) + (1 << 20), dtype=torch.uint8, device="cuda")
    base = (big.data_ptr() + 255) // 256 * 256
    src_t = D.synthetic((n,), 2)
    # the input lives at the start of the big block
    torch.cuda.synchronize()
    # device-to-device copy of the synthetic field into the block (torch copy: plumbing)
    big_f = big[base - big.data_ptr(): base - big.data_ptr() + nbytes].view(torch.float64)
    big_f.copy_(src_t)
    del src_t
    for off in offsets:
        dst = base + nbytes + off
        ms = time_it(lib, a.op, base, dst, shape, a.reps)
        print(json.dumps({"op": a.op, "placement": "one block", "out_minus_in_end_bytes": off, "in_mod_2MiB": base % (2 << 20), "out_mod_2MiB": dst % (2 << 20),
                          "ms": round(ms, 4), "frac_8TBps": round(alg / (ms * 1e-3) / 8e12, 4)}), flush=True)
    del big, big_f
    for r in range(a.fresh):
        torch.cuda.empty_cache()
        pad = torch.empty((r * 37 + 1) << 20, dtype=torch.uint8, device="cuda")  # shifts what the allocator hands out next
        x = D.synthetic(tuple(shape), 2)
        y = torch.empty(tuple(shape), dtype=torch.float64, device="cuda")
        ms = time_it(lib, a.op, x.data_ptr(), y.data_ptr(), shape, a.reps)
        print(json.dumps({"op": a.op, "placement": f"fresh allocations, round {r}", "in_ptr": hex(x.data_ptr()), "out_ptr": hex(y.data_ptr()),
                          "out_minus_in": y.data_ptr() - x.data_ptr(), "ms": round(ms, 4), "frac_8TBps": round(alg / (ms * 1e-3) / 8e12, 4)}), flush=True)
        del x, y, pad


if __name__ == "__main__":
    main()
