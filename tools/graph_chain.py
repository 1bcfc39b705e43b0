#!/usr/bin/env python3
"""A chain of Grid operators on a small grid: eager calls against one hipGraph replay (xgcm_amd.graphs.capture).

    python tools/graph_chain.py [--shape 4,320,256] [--reps 300]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from xgcm_amd import DataArray, Dataset, Grid  # noqa: E402
from xgcm_amd import device as D  # noqa: E402
from xgcm_amd.graphs import capture  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4,320,256")
    ap.add_argument("--reps", type=int, default=300)
    a = ap.parse_args()
    nz, ny, nx = (int(v) for v in a.shape.split(","))
    coords = {"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0, "YC": np.arange(ny) + 0.5, "YG": np.arange(ny) * 1.0,
              "Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0}
    met = lambda shape, seed: D.synthetic(shape, seed, 0, 1.0, 1.0)  # noqa: E731
    dv = {"dxC": DataArray(met((ny, nx), 31), ("YC", "XG")), "dyC": DataArray(met((ny, nx), 32), ("YG", "XC")), "drF": DataArray(met((nz,), 33), ("Z",))}
    grid = Grid(Dataset(dv, coords), coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, metrics={("X",): ["dxC"], ("Y",): ["dyC"], ("Z",): ["drF"]}, autoparse_metadata=False)
    T = DataArray(D.synthetic((nz, ny, nx), 2), ("Z", "YC", "XC"), name="T")
    chains = {
        "5 operators (derivative X, interp Y, cumsum Y, integrate Z, interp [X,Y])":
            lambda: (grid.derivative(T, "X"), grid.interp(T, "Y"), grid.cumsum(T, "Y"), grid.integrate(T, "Z"), grid.interp(T, ["X", "Y"])),
        "bench step (interp X, diff X, interp Y, diff Y)":
            lambda: (grid.interp(T, "X"), grid.diff(T, "X"), grid.interp(T, "Y"), grid.diff(T, "Y")),
    }
    for name, fn in chains.items():
        step = capture(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(json.dumps({"chain": name, "shape": [nz, ny, nx], "cells": nz * ny * nx, "graph_replay_us": round((t1 - t0) / a.reps * 1e6, 1),
                          "eager_us": round((t2 - t1) / a.reps * 1e6, 1)}))


if __name__ == "__main__":
    main()
