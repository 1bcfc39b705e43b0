"""A/B two builds of libxgcm_hip.so on the same GPU: time the headline stencils through the raw C ABI.

    python tools/abi_ab.py xgcm_amd/libxgcm_hip.so build/ab/lib_old.so

Prints one JSON line per (library, op): median / min ms over `--reps` launches of 75 x 2400 x 3600 f64.
Development aid only (used to tell a code regression from box-to-box variance)."""

import argparse
import ctypes as C
import json
import sys

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    shape = (75, 2400, 3600)
    x = torch.rand(shape, dtype=torch.float64, device="cuda")
    out = torch.empty_like(x)
    shp = (C.c_int64 * 3)(*shape)
    cases = [("interp_X periodic", 1, 2, 1), ("diff_X periodic", 0, 2, 1), ("interp_Y extend", 1, 1, 3),
             ("diff_Y extend", 0, 1, 3), ("diff_Z periodic", 0, 0, 1)]
    libs = [(p, C.CDLL(p)) for p in a.libs]
    for rnd in range(a.rounds):
        for path, lib in libs:
            fn = lib.xg_stencil1d_f64
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                           C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            for name, op, axis, bc in cases:
                def call():
                    rc = fn(op, x.data_ptr(), out.data_ptr(), shp, 3, axis, shape[axis], 1, 0, bc, 0.0, None, None, None,
                            None, None)
                    assert rc == 0, rc
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
                # the library launches on the NULL stream when `stream` is NULL; events on torch's default stream
                # (also the NULL stream) bracket it
                ev[0].record()
                for i in range(a.reps):
                    call()
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps))
                print(json.dumps({"round": rnd, "lib": path, "op": name, "median_ms": round(ms[len(ms) // 2], 4),
                                  "min_ms": round(ms[0], 4), "frac_8TBps": round(x.numel() * 16 / (ms[len(ms) // 2] * 1e-3) / 8e12, 4)}),
                      flush=True)


if __name__ == "__main__":
    sys.exit(main())
