#!/usr/bin/env python3
"""Golden vectors for the vertical coordinate transform from the REFERENCE's own kernel bodies.

TEST INFRASTRUCTURE -- runs in the build container only (it reads /root/reference), never on the GPU box, never from
the product.  Writes tests/golden/transform_kernels_reference.npz.

The reference's two compute kernels (`xgcm/transform.py:15-41` `_interp_1d_linear`, `:88-142`
`_interp_1d_conservative`) are numba gufuncs, and numba is not installable here.  Their BODIES are plain Python over
numpy scalars and `np.interp`, so this script loads the reference's `transform.py` FILE unmodified with two stand-in
modules -- `numba`, whose `guvectorize` runs the decorated body column by column over the broadcast loop dims (what the
gufunc machinery does; per-element arithmetic stays in the array's dtype because numpy scalars keep it), and `xarray`
(only named at import time by the mid-level functions, which are not called) -- and records what the reference's own
`interp_1d_linear` / `interp_1d_conservative` return on the seeded hard columns of tests/test_transform.py
(increasing / decreasing / duplicates / NaN head, tail, holes / non-monotonic; float64 and float32; mask_edges,
bypass_checks, logarithmic; increasing and decreasing bins).
`tests/golden/transform_cases.json` (the reference's own 24-case test table) stays as the other pin.

    python oracle/make_golden_transform.py
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference/xgcm/transform.py"


class _Type:
    def __getitem__(self, _):
        return self


def guvectorize(signatures, layout, **_kw):
    """Run the decorated body once per column of the broadcast loop dims, like the gufunc it would compile to."""
    import re

    ins, outs = layout.split("->")
    core_in = [tuple(c for c in part.split(",") if c) for part in re.findall(r"\(([^)]*)\)", ins)]
    core_out = tuple(c for c in re.findall(r"\(([^)]*)\)", outs)[0].split(",") if c)

    def deco(body):
        def gufunc(*args):
            arrs = [np.asarray(a) for a in args]
            dtype = np.result_type(*[a.dtype for a, c in zip(arrs, core_in) if c])
            arrs = [a.astype(dtype) if c else a for a, c in zip(arrs, core_in)]  # numba picks ONE signature: all f64 or all f32
            loops = [a.shape[:a.ndim - len(c)] for a, c in zip(arrs, core_in)]
            loop_shape = np.broadcast_shapes(*loops)
            sizes = {}
            for a, c in zip(arrs, core_in):
                for name, n in zip(c, a.shape[a.ndim - len(c):]):
                    assert sizes.setdefault(name, n) == n
            out = np.empty(loop_shape + tuple(sizes[c] for c in core_out), dtype=dtype)
            views = [np.broadcast_to(a, loop_shape + a.shape[a.ndim - len(c):]) if c else a for a, c in zip(arrs, core_in)]
            for idx in np.ndindex(*loop_shape):
                call = [(v[idx] if c else (v.item() if v.ndim == 0 else v[idx])) for v, c in zip(views, core_in)]
                body(*call, out[idx])
            return out

        gufunc.__wrapped__ = body
        return gufunc

    return deco


def load_reference_transform():
    nb = types.ModuleType("numba")
    nb.boolean = nb.float32 = nb.float64 = _Type()
    nb.guvectorize = guvectorize
    xr = types.ModuleType("xarray")
    xr.DataArray = type("DataArray", (), {})
    xr.Dataset = type("Dataset", (), {})
    saved = {k: sys.modules.get(k) for k in ("numba", "xarray")}
    sys.modules.update({"numba": nb, "xarray": xr})
    try:
        spec = importlib.util.spec_from_file_location("reference_transform", REFERENCE)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                del sys.modules[k]
            else:
                sys.modules[k] = v
    return mod


def main():
    from oracle import refimpl as R
    from tests import test_transform as TT

    P = load_reference_transform()
    out = {}
    kinds = ["increasing", "decreasing", "duplicates", "nan_tail", "nan_head", "nan_holes", "nonmonotonic"]
    shape = (3, 5, 17)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for dtype in (np.float64, np.float32):
            dn = np.dtype(dtype).name
            phi = (R.synthetic_field(shape, 5) * 10).astype(dtype)
            phi[0, 0, 4] = np.nan
            out[f"{dn}/phi"] = phi
            for kind in kinds:
                theta = TT._columns(shape, 3, kind).astype(dtype)
                out[f"{dn}/{kind}/theta"] = theta
                levels = np.concatenate([[-1.0, np.nan], np.linspace(0.0, float(np.nanmax(theta)) + 1, 23), [theta[1, 2, 5]]]).astype(dtype)
                clean = np.sort(np.concatenate([levels[~np.isnan(levels)], [theta[0, 0, 0], theta[2, 4, -1], theta[1, 2, 5]],
                                                np.nan_to_num(theta[2, 3, 6:9], nan=1.0)])).astype(dtype)
                for ln, lv in (("levels", levels), ("clean", clean)):
                    out[f"{dn}/{kind}/{ln}"] = lv
                    for mask in (False, True):
                        for bypass in (False, True):
                            key = f"{dn}/{kind}/{ln}/linear/m{int(mask)}b{int(bypass)}"
                            try:
                                out[key] = P.interp_1d_linear(phi, theta, lv, mask_edges=mask, bypass_checks=bypass)
                            except (IndexError, ValueError):  # an all-NaN column: numba indexes an empty array (undefined there)
                                pass
                pos = np.abs(theta) + dtype(0.5)
                lvl = (np.abs(levels[2:]) + dtype(0.25)).astype(dtype)
                out[f"{dn}/{kind}/log_theta"] = pos
                out[f"{dn}/{kind}/log_levels"] = lvl
                try:
                    out[f"{dn}/{kind}/log/linear"] = P.interp_1d_linear(phi, pos, lvl, mask_edges=True, logarithmic=True)
                except (IndexError, ValueError):
                    pass
                # conservative: theta on the n + 1 cell vertices, increasing and decreasing bins
                theta_o = np.concatenate([theta[..., :1] - dtype(0.7), theta], axis=-1).astype(dtype)
                out[f"{dn}/{kind}/theta_outer"] = theta_o
                hi = float(np.nanmax(theta_o))
                for bn, bins in (("inc", np.linspace(-0.5, hi + 0.5, 12)), ("dec", np.linspace(hi + 0.5, -0.5, 9)),
                                 ("fine", np.linspace(0.0, hi, 41))):
                    bins = bins.astype(dtype)
                    out[f"{dn}/{kind}/bins_{bn}"] = bins
                    out[f"{dn}/{kind}/conservative/{bn}"] = P.interp_1d_conservative(phi, theta_o, bins)
    for k, v in out.items():
        assert v.dtype in (np.float64, np.float32), (k, v.dtype)
    path = os.path.join(ROOT, "tests", "golden", "transform_kernels_reference.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)} arrays, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
