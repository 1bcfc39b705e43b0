#!/usr/bin/env python3
"""Rates of the integer paths (int64 / int32 lanes, xg_convert around the narrower dtypes) next to their float64 twins, through the device layer
at full size: what exactness costs.  One JSON line per (dtype, operator)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from xgcm_amd import device as D  # noqa: E402


def timeit(fn, reps=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def main():
    nz, ny, nx = 75, 2400, 3600
    cells = nz * ny * nx
    f = D.synthetic((nz, ny, nx), 2)
    base = {"float64": f}
    # integer fields of every width from the same bits (reinterpreted / converted in HBM)
    base["int64"] = f.view(torch.int64)
    base["uint64"] = f.view(torch.uint64)
    base["int32"] = D.convert(base["int64"], np.int32)
    base["uint32"] = base["int32"].view(torch.uint32)
    base["int16"] = D.convert(base["int64"], np.int16)
    base["uint8"] = D.convert(base["int64"], np.uint8)
    ops = [("diff X", lambda t: D.stencil1d("diff", t, 2, 1, 0, "periodic")),
           ("interp X", lambda t: D.stencil1d("interp", t, 2, 1, 0, "periodic")),
           ("max Y", lambda t: D.stencil1d("max", t, 1, 1, 0, "extend")),
           ("diff Z", lambda t: D.stencil1d("diff", t, 0, 1, 0, "fill", 3)),
           ("cumsum Y", lambda t: D.cumsum1d(t, 1, 0, 1, 1, 0, "fill", 0)),
           ("cumsum Z", lambda t: D.cumsum1d(t, 0, 0, 1, 1, 0, "fill", 0)),
           ("sum Z", lambda t: D.reduce1d(t, 0, None, True))]
    for name, t in base.items():
        esz = t.element_size()
        for op, fn in ops:
            ms = timeit(lambda: fn(t))
            out = fn(t)
            moved = cells * esz + out.numel() * out.element_size()  # bytes the caller sees: storage dtype in, result dtype out
            print(json.dumps({"dtype": name, "op": op, "ms": round(ms, 3), "result_dtype": str(out.dtype).replace("torch.", ""),
                              "caller_GBps": round(moved / ms / 1e6, 1), "vs_8TBps_on_caller_bytes": round(moved / ms / 1e6 / 8000, 3)}), flush=True)
            del out


if __name__ == "__main__":
    main()
