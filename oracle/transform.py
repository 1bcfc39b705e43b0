"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the vertical coordinate transform.

Restates the two numba gufuncs of the reference and their thin numpy wrappers:

  * `_interp_1d_linear`      xgcm/transform.py:15-41   + wrapper `interp_1d_linear`      :44-86
  * `_interp_1d_conservative` xgcm/transform.py:88-142 + wrapper `interp_1d_conservative` :145-193

The kernels live in a third-party-compiled form in the reference: **numba** (`pyproject.toml:136`,
`numba = "*"`, unpinned; absent from this image) JIT-compiles them.  The arithmetic they contain
is restated here with numpy (present): the only library routine inside is `np.interp`, which numba
implements as a port of numpy's C routine (same search, same slope formula, float64 arithmetic
whatever the input type), so this oracle calls numpy's own `np.interp` per column.

Pinning status
--------------
* the `cases` table of the reference's test-suite (xgcm/test/test_transform.py:40-686: inputs and
  expected outputs, 24 cases) is evaluated in the build container by oracle/make_golden.py and
  committed as tests/golden/transform_cases.json;
* the BODIES of the two gufuncs are plain Python over numpy scalars and `np.interp`: oracle/make_golden_transform.py
  loads the reference's transform.py unmodified with a stand-in `numba` whose `guvectorize` runs the body column by
  column, and records what `interp_1d_linear` / `interp_1d_conservative` return on the seeded hard columns of
  tests/test_transform.py (NaN head / tail / holes, duplicates, decreasing, non-monotonic; float64 and float32; every
  mask_edges / bypass_checks combination, logarithmic, increasing / decreasing / fine bins) ->
  tests/golden/transform_kernels_reference.npz, tests/test_transform.py::test_transform_kernels_equal_reference_kernel_outputs
  (this oracle AND the product, bit-exact; float32 log within the stated tolerance);
* the low-level property tests of the same file (:849-921) are re-created in tests/test_transform.py;
* parity unpinned: behaviour on an all-NaN theta column (numba indexes an empty array there,
  undefined); here such a column is neither flipped nor masked.
* deviation: a decreasing conservative target on N-D data is un-flipped along the bin axis (the
  reference flips axis 0 of the result, transform.py:190-192, which is the bin axis for 1-D only).
"""

from __future__ import annotations

import numpy as np


def interp_1d_linear(phi, theta, target_theta_levels, mask_edges=False, bypass_checks=False, logarithmic=False):
    """phi, theta: (..., n) broadcastable; target: (..., m) broadcastable -> (..., m)."""
    phi = np.asarray(phi)
    theta = np.asarray(theta)
    target = np.asarray(target_theta_levels)
    if logarithmic:  # transform.py:83-85 (the input dtype's own log)
        with np.errstate(all="ignore"):
            theta = np.log(theta)
            target = np.log(target)
    lead = np.broadcast_shapes(phi.shape[:-1], theta.shape[:-1], target.shape[:-1])
    phi_b = np.broadcast_to(phi, lead + phi.shape[-1:])
    theta_b = np.broadcast_to(theta, lead + theta.shape[-1:])
    target_b = np.broadcast_to(target, lead + target.shape[-1:])
    out = np.empty(lead + target.shape[-1:], dtype=np.result_type(phi.dtype, np.float32))
    for idx in np.ndindex(*lead):
        p, t, lev = phi_b[idx], theta_b[idx], target_b[idx]
        if not bypass_checks:  # transform.py:27-31
            valid = t[~np.isnan(t)]
            if valid.size and valid[-1] < valid[0]:
                t = t[::-1]
                p = p[::-1]
        col = np.interp(lev, t, p)  # transform.py:33
        if mask_edges:  # transform.py:35-41
            if np.any(~np.isnan(t)):
                tmax, tmin = np.nanmax(t), np.nanmin(t)
                col = np.where((lev < tmin) | (lev > tmax), np.nan, col)
        out[idx] = col
    return out


def _conservative_column(phi, theta_1, theta_2, theta_hat_1, theta_hat_2):
    """transform.py:98-142, one column, in the storage dtype (numba keeps float32 arithmetic)."""
    dt = phi.dtype.type
    n, m = len(theta_1), len(theta_hat_1)
    output = np.full(m, np.nan, dtype=phi.dtype)
    for i in range(n):
        t1, t2 = theta_1[i], theta_2[i]
        if np.isnan(t1) and np.isnan(t2):
            continue
        elif np.isnan(t1):
            theta_min = theta_max = t2
        elif np.isnan(t2):
            theta_min = theta_max = t1
        elif t1 < t2:
            theta_min, theta_max = t1, t2
        else:
            theta_min, theta_max = t2, t1
        for j in range(m):
            if np.isnan(phi[i]):
                continue
            if (theta_hat_1[j] > theta_max) or (theta_hat_2[j] < theta_min):
                pass
            elif theta_max == theta_min:
                output[j] = phi[i] if np.isnan(output[j]) else dt(output[j] + phi[i])
            else:
                theta_hat_min = max(theta_min, theta_hat_1[j])
                theta_hat_max = min(theta_max, theta_hat_2[j])
                alpha = dt(dt(theta_hat_max - theta_hat_min) / dt(theta_max - theta_min))
                add = dt(alpha * phi[i])
                output[j] = add if np.isnan(output[j]) else dt(output[j] + add)
    return output


def interp_1d_conservative(phi, theta, target_theta_bins):
    """phi: (..., n) extensive cell values; theta: (..., n+1) vertex values; bins: (m,) -> (..., m-1)."""
    phi = np.asarray(phi)
    theta = np.asarray(theta)
    bins = np.asarray(target_theta_bins)
    assert phi.shape[-1] == theta.shape[-1] - 1
    assert bins.ndim == 1
    d = np.diff(bins)
    if np.all(d < 0):  # transform.py:174-181
        flip = True
        bins = bins[::-1]
    elif np.all(d > 0):
        flip = False
    else:
        raise ValueError("Target values are not monotonic")
    lead = np.broadcast_shapes(phi.shape[:-1], theta.shape[:-1])
    phi_b = np.broadcast_to(phi, lead + phi.shape[-1:])
    theta_b = np.broadcast_to(theta, lead + theta.shape[-1:])
    bins = bins.astype(phi.dtype, copy=False) if phi.dtype == np.float32 and theta.dtype == np.float32 else bins
    out = np.empty(lead + (len(bins) - 1,), dtype=phi_b.dtype if phi_b.dtype.kind == "f" else np.float64)
    for idx in np.ndindex(*lead):
        p = np.asarray(phi_b[idx], dtype=out.dtype)
        t = np.asarray(theta_b[idx], dtype=out.dtype)
        out[idx] = _conservative_column(p, t[:-1], t[1:], bins[:-1].astype(out.dtype), bins[1:].astype(out.dtype))
    if flip:
        # transform.py:190-192 writes `out[::-1]`, i.e. axis 0 of the result -- the bin axis only for
        # 1-D input (all the reference tests).  The evident intent (undo the flip of the bins) is the
        # bin axis; product and oracle both do that (deviation).
        out = out[..., ::-1]
    return out
