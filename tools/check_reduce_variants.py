#!/usr/bin/env python3
"""Results of the contiguous-axis reduction variants (xg_set_tunable reduce_wg) against the default kernels, on a GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xgcm_amd import _hip
from xgcm_amd import device as D

lib = _hip.load()
def setv(v):
    assert lib.xg_set_tunable(b"reduce_wg", int(v)) == 0
bad = 0
for shape in [(75, 240, 3600), (7, 33, 1024), (5, 40, 1026), (3, 17, 4098), (2, 3, 600), (9, 64, 520)]:
    T = D.synthetic(shape, 2)
    T.view(-1)[::977] = float("nan")
    w = D.synthetic((1,) + shape[1:], 31, 0, 1000.0, 1000.0)
    w3 = D.synthetic(shape, 32, 0, 1000.0, 1000.0)
    for wname, wt in (("none", None), ("w(Y,X)", w), ("w(Z,Y,X)", w3)):
        for mode in (True, False, "valid", "all", "mean_valid", "mean_all", "pair_valid"):
            if wt is None and mode not in (True, False):
                continue
            setv(0)
            ref = D.reduce1d(T, 2, wt, mode).double().cpu().numpy()
            for v in (1, 3, 5, 8, 16, 9, 17):
                setv(v)
                got = D.reduce1d(T, 2, wt, mode).double().cpu().numpy()
                # re-associated sums: the error is relative to the sum of the |terms| (a row of mean-zero values cancels)
                scale = float(np.nanmax(np.abs(ref))) if np.isfinite(np.nanmax(np.abs(ref))) else 1.0
                ok = np.allclose(got, ref, rtol=1e-12, atol=1e-12 * scale, equal_nan=True)
                if not ok:
                    bad += 1
                    print("MISMATCH", shape, wname, mode, v, np.nanmax(np.abs(got - ref) / np.abs(ref)))
setv(0)
print("checked; mismatches:", bad)
sys.exit(1 if bad else 0)
