#!/usr/bin/env python3
"""Survey of the operator surface at full size on one GPU: every Grid operator x axis x metric pattern that a
MITgcm-like C-grid offers, timed through the public API (median of `--reps` launches after a warm-up), with the
algorithmic bytes of the FUSED form.  Its purpose is to find cliffs -- combinations that fall far below their
siblings -- not to produce the judged numbers (bench.py, tools/bench_configs.py).

    python tools/survey.py [--reps 5] [--shape 75,2400,3600]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shape", default="75,2400,3600")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    a = ap.parse_args()
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
    for name, fn, bpc in cases:
        try:
            ms = timeit(fn, a.reps)
        except Exception as exc:  # noqa: BLE001
            print(json.dumps({"op": name, "error": f"{type(exc).__name__}: {exc}"[:200]}), flush=True)
            continue
        bpc = bpc * esz / 8.0  # the byte counts above are written for 8-byte elements
        gbs = cells * bpc / (ms * 1e-3) / 1e9
        print(json.dumps({"op": name, "dtype": a.dtype, "ms": round(ms, 3), "bytes_per_cell_fused": round(bpc, 3), "GBps": round(gbs, 1),
                          "frac_8TBps": round(gbs / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
