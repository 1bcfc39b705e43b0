#!/usr/bin/env python3
"""Soak test of the chained scans / reductions (K5c, K4c, K4cz): hundreds of launches over eight shapes, both directions, all
boundary modes, with and without a shared metric, each compared bit for bit with the marching kernel.

    python tools/soak_chain.py      # 400 iterations x 3 chained launches in about 2 s on one MI355X
"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from xgcm_amd import device as D, _hip
rng = np.random.default_rng(5)
t0 = time.time(); n_ok = 0
shapes = [(75, 2400, 3600), (8, 4096, 512), (300, 64, 3600), (3, 257, 130), (1, 1000, 64), (40, 333, 1026), (600, 128, 256), (17, 65, 66)]
big = D.synthetic((75, 2400, 3600), 3)
ref_big = None
for it in range(400):
    shp = shapes[it % len(shapes)]
    if shp == (75, 2400, 3600):
        a = big
    else:
        a = D.synthetic(shp, 100 + it)
    rev = bool(it & 1); bc = ["fill", "extend", "periodic"][it % 3]
    w = D.synthetic((1,) + shp[1:], 7, 0, 1.0, 1.0) if it % 4 == 0 else None
    _hip.set_tunable("scan_chain", 0)
    want = D.cumsum1d(a, 1, 0, 1, 1, 0, bc, 0.5, rev, True, w, None)
    wr = D.reduce1d(a, 1, w, "mean_valid") if w is not None else None
    _hip.set_tunable("scan_chain", 2)
    for rep in range(3):
        got = D.cumsum1d(a, 1, 0, 1, 1, 0, bc, 0.5, rev, True, w, None)
        assert torch.equal(got, want) or torch.allclose(got, want, rtol=0, atol=0, equal_nan=True), (it, shp)
        if w is not None:
            gr = D.reduce1d(a, 1, w, "mean_valid")
            assert torch.equal(gr, wr) or torch.allclose(gr, wr, rtol=0, atol=0, equal_nan=True), (it, shp, "reduce")
    n_ok += 1
torch.cuda.synchronize()
print(f"soak: {n_ok} iterations x 3 chained launches (+ weighted reductions) identical to the march, {time.time()-t0:.1f} s")
