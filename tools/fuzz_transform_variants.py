"""On-demand robustness run (not part of the suites): every way the transforms get their outputs to memory (direct stores,
level table in LDS, whole-column tile, rings of 4 / 8 / 16 rows; one accumulator window per lane or per wave of 4 / 8 / 16
bins) on 120 random shapes / level counts / defective columns, compared bit for bit."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgcm_amd import device as D, _hip
rng = np.random.default_rng(77)
t0 = time.time(); n = 0
def same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.25), torch.nan_to_num(b, nan=-7.25))
for it in range(120):
    nz = int(rng.integers(2, 45)); ny = int(rng.integers(1, 9)); nx = int(rng.choice([1, 3, 64, 65, 130, 257, 512]))
    m = int(rng.integers(1, 71))
    dtype = np.float64 if it % 3 else np.float32
    theta = np.cumsum(rng.random((nz, ny, nx)) * 2 + 0.01, axis=0)
    k = rng.integers(0, 5)
    if k == 1: theta[rng.integers(0, nz), rng.integers(0, ny), rng.integers(0, nx)] = np.nan
    if k == 2: theta[:, 0, : max(1, nx // 3)] = theta[::-1, 0, : max(1, nx // 3)]
    if k == 3: theta[nz // 2:, :, ::7] -= 5.0
    if k == 4: theta[:, :, ::5] *= 0.02
    theta = theta.astype(dtype)
    phi = rng.standard_normal((nz, ny, nx)).astype(dtype)
    if it % 4 == 0: phi[rng.integers(0, nz), 0, 0] = np.nan
    levels = np.sort(rng.random(m) * float(np.nanmax(theta)) * 1.1).astype(dtype).reshape(m, 1, 1)
    edges = np.linspace(0.0, float(np.nanmax(theta)) * 1.05, m + 1).astype(dtype)
    theta_o = np.concatenate([theta[:1] - dtype(1.0), theta], 0)
    _hip.set_tunable("transform_stage", 0); _hip.set_tunable("transform_win", 2)
    lin0 = D.transform_linear(phi, theta, levels, 0, mask_edges=bool(it & 1), bypass_checks=bool(it & 8))
    con0 = D.transform_conservative(phi, theta_o, edges, 0)
    for stage, ring, win, cwin, lean in ((3, 4, 1, 4, 3), (3, 8, 1, 8, 3), (3, 16, 1, 16, 3), (3, 8, 1, 8, 0), (3, 4, 1, 16, 0), (2, 8, 1, 8, 3), (1, 8, 1, 8, 3)):
        _hip.set_tunable("transform_lean", lean)  # bit 0: lean linear loop, bit 1: lean conservative kernel (K9e)
        _hip.set_tunable("transform_stage", stage); _hip.set_tunable("transform_ring", ring)
        _hip.set_tunable("transform_win", win); _hip.set_tunable("transform_cwin", cwin)
        lin = D.transform_linear(phi, theta, levels, 0, mask_edges=bool(it & 1), bypass_checks=bool(it & 8))
        con = D.transform_conservative(phi, theta_o, edges, 0)
        assert same(lin, lin0), (it, "linear", stage, ring, lean, (nz, ny, nx), m, k)
        assert same(con, con0), (it, "conservative", win, cwin, lean, (nz, ny, nx), m, k)
        n += 2
print(f"{n} transform variant comparisons identical, {time.time()-t0:.0f} s")
