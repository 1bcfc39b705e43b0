#!/usr/bin/env python3
"""Survey of the operator surface at full size on one GPU: every Grid operator x axis x metric pattern that a
MITgcm-like C-grid offers, timed through the public API (median of `--reps` launches after a warm-up), with the
algorithmic bytes of the FUSED form.  Its purpose is to find cliffs -- combinations that fall far below their
siblings -- not to produce the judged numbers (bench.py, tools/bench_configs.py).

    python tools/survey.py [--reps 5] [--shape 75,2400,3600]
    python tools/survey.py --trace          # the same run under `rocprofv3 --kernel-trace`: kernel time per call next to
                                            # the event-timed wall time of the SAME process

`--trace` answers "is a slow operator slow in its kernels or around them": `trace_us` is the summed duration of every kernel
the operator launches per call (profiler's clock), `ms` the HIP-event time per call of the same process.  When two tools
disagree on one operator (r03end: cumsum Z 1.749 ms in roofline_table.py, 1.91 ms here) the pair tells whether the path
(extra launches, allocation) or the state of the device (what ran before, for how long) makes the difference.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from xgcm_amd import DataArray, Dataset, Grid  # noqa: E402
from xgcm_amd import device as D  # noqa: E402


def timeit(fn, reps):
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


def traced(argv):
    """re-run this script with --mark under rocprofv3 --kernel-trace and join the kernel durations to its JSON lines"""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="survey_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--mark"] + argv
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        print(json.dumps({"error": r.stdout[-600:]}), flush=True)
        return
    con = sqlite3.connect(dbs[0])
    blocks, cur = {}, None
    for name, dur, gx in con.execute("select name, duration, grid_x from kernels order by start"):
        k = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if k.startswith("k_fill_synthetic") and gx % 4096 == 0 and gx < 4096 * 512:
            cur = gx // 4096 - 1
            continue
        if cur is not None:
            e = blocks.setdefault(cur, {})
            e.setdefault(k, []).append(dur / 1e3)
    shutil.rmtree(tmp, ignore_errors=True)
    for i, ln in enumerate(lines):
        if "error" in ln:
            print(json.dumps(ln), flush=True)
            continue
        kern = blocks.get(ln.get("index", i), {})
        calls = ln.get("calls", 1)
        # per call: every kernel's launches of the block / calls, each at its median duration (the first call runs on cold caches)
        per_call = 0.0
        detail = {}
        for k, ds in kern.items():
            ds = sorted(ds)
            med = ds[len(ds) // 2]
            n = len(ds) / calls
            per_call += med * n
            if med * n > 20:
                detail[k[:60]] = {"us": round(med, 1), "launches_per_call": round(n, 2)}
        ln["trace_us_per_call"] = round(per_call, 1)
        ln["wall_over_trace"] = round(ln["ms"] * 1e3 / per_call, 4) if per_call else None
        ln["kernels"] = detail
        print(json.dumps(ln), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shape", default="75,2400,3600")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--placements", type=int, default=3, help="fresh placements of the field (allocator cache emptied, allocations "
                    "shifted) each operator is timed on; `ms` is the median over them, `ms_min` / `ms_max` the spread")
    ap.add_argument("--only", default="", help="comma-separated substrings: run the operators whose name contains one of them")
    ap.add_argument("--mark", action="store_true", help="a marker dispatch (k_fill_synthetic of 4096 * (1 + index) cells) before every operator")
    ap.add_argument("--trace", action="store_true", help="run under rocprofv3 --kernel-trace and print kernel time per call next to the wall time")
    a = ap.parse_args()
    if a.trace:
        traced([x for x in sys.argv[1:] if x != "--trace"])
        return
    nz, ny, nx = (int(v) for v in a.shape.split(","))
    cells = nz * ny * nx
    tdt = torch.float32 if a.dtype == "f32" else torch.float64
    esz = 4 if a.dtype == "f32" else 8
    met = lambda shape, seed: D.synthetic(shape, seed, 0, 1000.0, 1000.0, dtype=tdt)  # noqa: E731
    coords = {"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0, "YC": np.arange(ny) + 0.5, "YG": np.arange(ny) * 1.0,
              "Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0}
    dv = {"dxC": DataArray(met((ny, nx), 31), ("YC", "XG")), "dxT": DataArray(met((ny, nx), 35), ("YC", "XC")),
          "dyC": DataArray(met((ny, nx), 32), ("YG", "XC")), "dyT": DataArray(met((ny, nx), 36), ("YC", "XC")),
          "drF": DataArray(met((nz,), 33), ("Z",)), "drC": DataArray(met((nz,), 34), ("Zl",)),
          "rA": DataArray(met((ny, nx), 37), ("YC", "XC")), "rAz": DataArray(met((ny, nx), 38), ("YG", "XG")),
          "hFacC": DataArray(met((nz, ny, nx), 39), ("Z", "YC", "XC"))}
    grid = Grid(Dataset(dv, coords),
                coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"},
                metrics={("X",): ["dxC", "dxT"], ("Y",): ["dyC", "dyT"], ("Z",): ["drF", "drC"], ("X", "Y"): ["rA", "rAz"]},
                autoparse_metadata=False)
    T = DataArray(D.synthetic((nz, ny, nx), 2, dtype=tdt), ("Z", "YC", "XC"), name="T")
    shift = [None]

    def replace_field(k):
        """the field on a fresh allocation (the lambdas below read `T.data` at call time)"""
        T.data = None
        shift[0] = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        shift[0] = torch.empty((k * 37 + 1) << 20, dtype=torch.uint8, device="cuda")
        T.data = D.synthetic((nz, ny, nx), 2, dtype=tdt)

    mB = 8.0 / nz  # bytes per cell of a 2-D metric read once
    cases = []
    for ax in "XYZ":
        m2 = mB if ax in "XY" else 0.0
        cases += [(f"diff {ax}", lambda ax=ax: grid.diff(T, ax), 16), (f"interp {ax}", lambda ax=ax: grid.interp(T, ax), 16),
                  (f"min {ax}", lambda ax=ax: grid.min(T, ax), 16),
                  (f"derivative {ax}", lambda ax=ax: grid.derivative(T, ax), 16 + m2),
                  (f"interp {ax} metric_weighted", lambda ax=ax: grid.interp(T, ax, metric_weighted=ax), 16 + 2 * m2),
                  (f"cumsum {ax}", lambda ax=ax: grid.cumsum(T, ax), 16),
                  (f"cumint {ax}", lambda ax=ax: grid.cumint(T, ax), 16 + m2),
                  (f"integrate {ax}", lambda ax=ax: grid.integrate(T, ax), 8 + m2),
                  (f"average {ax}", lambda ax=ax: grid.average(T, ax), 8 + m2)]
    cases += [("integrate [X,Y] (area)", lambda: grid.integrate(T, ["X", "Y"]), 8 + mB),
              ("average [X,Y] (area)", lambda: grid.average(T, ["X", "Y"]), 8 + mB),
              ("integrate [X,Y,Z] (volume)", lambda: grid.integrate(T, ["X", "Y", "Z"]), 8 + mB),
              ("interp [X,Y]", lambda: grid.interp(T, ["X", "Y"]), 16),
              ("diff [Y,X]", lambda: grid.diff(T, ["Y", "X"]), 16),
              ("interp [X,Y] metric_weighted (X,Y)", lambda: grid.interp(T, ["X", "Y"], metric_weighted=("X", "Y")), 16 + 2 * mB),
              ("T * hFacC (3-D metric)", lambda: T * dv["hFacC"], 24), ("T / dxT (2-D)", lambda: T / dv["dxT"], 16 + mB)]
    for index, (name, fn, bpc) in enumerate(cases):
        if a.only and not any(w.strip() and w.strip() in name for w in a.only.split(",")):
            continue
        try:
            if a.mark:
                D.synthetic((4096 * (1 + index),), 1)
            per = []
            for k in range(max(1, a.placements)):
                if k > 0:
                    replace_field(k)
                per.append(timeit(fn, a.reps))
            per.sort()
            ms = per[len(per) // 2]
        except Exception as exc:  # noqa: BLE001
            print(json.dumps({"op": name, "error": f"{type(exc).__name__}: {exc}"[:200]}), flush=True)
            continue
        bpc = bpc * esz / 8.0  # the byte counts above are written for 8-byte elements
        gbs = cells * bpc / (ms * 1e-3) / 1e9
        print(json.dumps({"op": name, "index": index, "calls": (a.reps + 2) * max(1, a.placements), "dtype": a.dtype, "ms": round(ms, 3),
                          "ms_min": round(per[0], 3), "ms_max": round(per[-1], 3), "bytes_per_cell_fused": round(bpc, 3), "GBps": round(gbs, 1),
                          "frac_8TBps": round(gbs / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
