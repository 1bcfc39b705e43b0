#!/usr/bin/env python3
"""The code blocks of the reference's documentation, executed against the reference AND against `xgcm_amd`, compared.

TEST INFRASTRUCTURE -- build container only (reads /root/reference/docs at run time; nothing is copied into the repo).

    python oracle/run_reference_docs.py [--fused] [--backend oracle-double|host-abi] [-v]

The reference's user guide is executable markdown (`docs/*.md`, "execute: true"): grids, boundary conditions, grid ufuncs,
the vector-calculus examples (divergence / gradient / vorticity written as user grid ufuncs), grid topology -- and the
`Grid.transform` notebook (`docs/transform.ipynb`: linear and conservative transforms of a profile, sigma -> pressure levels,
the analytic 3-D atmosphere and derivatives on its isobaric grid; its cells that fetch CMIP6 / ROMS data over the network
raise the same `ModuleNotFoundError` / `NameError` in both runs) and the metrics notebook (`docs/grid_metrics.ipynb`:
integrate / average / cumint / derivative / metric-weighted interp; its download cell replaced by a synthetic dataset of the
same variables).  For every page
the python blocks run in order in ONE namespace, twice, in two child processes: `import xgcm` is the reference's package in
the first and a shim over `xgcm_amd` in the second (same trick as `oracle/run_reference_suite.py`); `xarray` is the real
package where importable, else the stand-in (`oracle/xr_min.py` + `xr_suite.py`); plotting is stubbed.  After each block
every labelled array in the namespace is snapshotted (dims, name, coordinate names, values) -- the parent compares the two
runs block by block: same variables, same snapshots (1e-12 for re-associated scans), same exception class where a block
raises.  `tests/test_reference_suite_live.py::test_documentation_examples` runs it wherever the reference tree is.
"""
import argparse
import json
import os
import pickle
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("XGCM_REFERENCE", "/root/reference")
PAGES = ["grids.md", "boundary_conditions.md", "grid_ufuncs.md", "ufunc_examples.md", "grid_topology.md", "transform.ipynb", "grid_metrics.ipynb"]


# `docs/grid_metrics.ipynb` opens a MITgcm example file fetched from zenodo (its cell 1); there is no network here, so that ONE
# cell is replaced by a synthetic dataset of the same variables, dims and COMODO attributes (own text; nothing of the file)
_MITGCM_LIKE = """
_rng = np.random.default_rng(0)
_nz, _ny, _nx = 6, 8, 10
_c = {
    "XC": ("XC", np.arange(_nx) + 0.5, {"axis": "X"}),
    "XG": ("XG", np.arange(_nx) * 1.0, {"axis": "X", "c_grid_axis_shift": -0.5}),
    "YC": ("YC", np.arange(_ny) + 0.5, {"axis": "Y"}),
    "YG": ("YG", np.arange(_ny) * 1.0, {"axis": "Y", "c_grid_axis_shift": -0.5}),
    "Z": ("Z", -(np.arange(_nz) + 0.5), {"axis": "Z"}),
    "Zl": ("Zl", -np.arange(_nz) * 1.0, {"axis": "Z", "c_grid_axis_shift": -0.5}),
    "time": ("time", np.arange(1) * 1.0, {"axis": "T"}),
}
def _f(*dims):
    n = {"time": 1, "Z": _nz, "Zl": _nz, "YC": _ny, "YG": _ny, "XC": _nx, "XG": _nx}
    return (dims, _rng.random([n[d] for d in dims]) + 0.5)
ds = xr.Dataset(
    {
        "UVEL": _f("time", "Z", "YC", "XG"), "VVEL": _f("time", "Z", "YG", "XC"),
        "THETA": _f("time", "Z", "YC", "XC"), "SALT": _f("time", "Z", "YC", "XC"),
        "hFacC": _f("Z", "YC", "XC"), "hFacW": _f("Z", "YC", "XG"), "hFacS": _f("Z", "YG", "XC"),
        "drF": _f("Z"), "dxC": _f("YC", "XG"), "dxG": _f("YG", "XC"), "dyC": _f("YG", "XC"), "dyG": _f("YC", "XG"),
        "rA": _f("YC", "XC"), "rAz": _f("YG", "XG"), "rAs": _f("YG", "XC"), "rAw": _f("YC", "XG"),
        "maskC": (("Z", "YC", "XC"), _rng.random((_nz, _ny, _nx)) > 0.2),
        "maskW": (("Z", "YC", "XG"), _rng.random((_nz, _ny, _nx)) > 0.2),
        "maskS": (("Z", "YG", "XC"), _rng.random((_nz, _ny, _nx)) > 0.2),
    },
    _c,
)
"""


def blocks_of(page):
    if page.endswith(".ipynb"):  # a notebook: its code cells, IPython magics dropped
        cells = json.load(open(os.path.join(REF, "docs", page)))["cells"]
        code = ["".join(ln for ln in c["source"] if not ln.lstrip().startswith("%")) for c in cells if c["cell_type"] == "code"]
        if page == "grid_metrics.ipynb":
            code = [_MITGCM_LIKE if "pooch.retrieve" in c else c for c in code]
        return code
    text = open(os.path.join(REF, "docs", page)).read()
    return [m.group(1) for m in re.finditer(r"^```python[^\n]*\n(.*?)^```", text, flags=re.S | re.M)]


# ---- child: run one implementation over every page --------------------------------------------------------------------
def child(which, backend, fused, out_path):
    import types
    import warnings

    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import fuzz_against_reference as F

    xr, RefGrid, OurGrid = F.load_both(backend)  # registers xarray (stand-in unless installed), numba stand-in, the device double
    if getattr(xr, "_is_xr_min", False):
        from oracle import xr_suite

        xr_suite.extend(xr)

    class _Plot:  # `.plot`, `.plot.quiver(...)`, `plt.figure()` ...: anything, returning itself
        def __call__(self, *a, **k):
            return self

        def __getattr__(self, name):
            return self

        def __iter__(self):
            return iter((self, self))

    for cls in (xr.DataArray, xr.Dataset):
        try:
            cls.plot = property(lambda self: _Plot())
        except (AttributeError, TypeError):
            pass
    mpl = types.ModuleType("matplotlib")
    plt = types.ModuleType("matplotlib.pyplot")
    plt.__getattr__ = lambda name: _Plot()

    def _subplots(nrows=1, ncols=1, **kw):  # `fig, (ax1, ax2, ax3) = plt.subplots(ncols=3)`
        n = int(nrows) * int(ncols)
        return _Plot(), (_Plot() if n == 1 else [_Plot() for _ in range(n)])

    plt.subplots = _subplots
    mpl.pyplot = plt
    sys.modules.setdefault("matplotlib", mpl)
    sys.modules.setdefault("matplotlib.pyplot", plt)

    loosened = []
    if which == "own":
        import xgcm_amd

        # `xr.testing.assert_equal(grid.integrate(u, ["X", "Y"]), (u * area).sum([...]))` in the metrics notebook: the product's
        # sums over two axes (and, on the GPU, along the contiguous axis) are re-associated -- equal to 1e-12, not to the bit
        # (DESIGN section 6).  The notebook's own assertion is held to that and every such case is counted in the report.
        _strict = xr.testing.assert_equal

        def _assert_equal(a, b):
            try:
                _strict(a, b)
            except AssertionError:
                xr.testing.assert_allclose(a, b, rtol=1e-12, atol=0.0)
                loosened.append(True)

        xr.testing.assert_equal = _assert_equal

        if fused:
            _init = xgcm_amd.Grid.__init__

            def _fused_init(self, *a, **k):
                k.setdefault("fuse", True)
                _init(self, *a, **k)

            xgcm_amd.Grid.__init__ = _fused_init
        shim = types.ModuleType("xgcm")
        for k in ("Grid", "Axis", "as_grid_ufunc", "apply_as_grid_ufunc", "GridUFunc"):
            setattr(shim, k, getattr(xgcm_amd, k))
        shim.__path__ = []
        for name, source in (("grid", "grid"), ("grid_ufunc", "grid_ufunc"), ("padding", "padding"), ("axis", "axis"),
                             ("metrics", "metrics"), ("gridops", "gridops"), ("metadata_parsers", "metadata"), ("sgrid", "metadata"),
                             ("comodo", "metadata"), ("transform", "transform")):
            mod = __import__(f"xgcm_amd.{source}", fromlist=["*"])
            sys.modules[f"xgcm.{name}"] = mod
            setattr(shim, name, mod)
        sys.modules["xgcm"] = shim

    def snap(v):
        if type(v).__name__ == "LazyArray":
            v = v.compute()
        if type(v).__name__ == "DataArray" and hasattr(v, "dims"):
            vals = np.asarray(v.values)
            return {"kind": "DataArray", "dims": tuple(v.dims), "name": v.name, "coords": sorted(v.coords),
                    "dtype": str(vals.dtype), "values": vals}
        if type(v).__name__ == "Dataset" and hasattr(v, "data_vars"):
            return {"kind": "Dataset", "vars": {k: snap(v[k]) for k in v.data_vars}}
        if isinstance(v, np.ndarray) and v.dtype.kind in "biuf" and v.size < 10 ** 6:
            return {"kind": "ndarray", "values": v}
        if isinstance(v, (tuple, list)) and v and all(type(x).__name__ in ("DataArray", "LazyArray") for x in v):
            return {"kind": "tuple", "items": [snap(x) for x in v]}
        if isinstance(v, dict) and v and all(type(x).__name__ in ("DataArray", "LazyArray") for x in v.values()):
            return {"kind": "dict", "items": {str(k): snap(x) for k, x in v.items()}}
        return None

    report = {}
    for page in PAGES:
        ns = {"__name__": "__docs__"}
        np.random.seed(0)
        per_block = []
        for i, code in enumerate(blocks_of(page)):
            entry = {"raised": None, "vars": {}}
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    exec(compile(code, f"{page}[{i}]", "exec"), ns)  # noqa: S102 -- the documentation's own code
            except (KeyboardInterrupt, SystemExit):
                raise
            except BaseException as exc:  # noqa: BLE001  (pytest's Skipped -- a block that chunks with dask -- is a BaseException)
                entry["raised"] = (type(exc).__name__, str(exc)[:200])
            for k, v in list(ns.items()):
                if k.startswith("_"):
                    continue
                try:
                    s = snap(v)
                except Exception as exc:  # noqa: BLE001
                    s = {"kind": "unreadable", "why": repr(exc)[:200]}
                if s is not None:
                    entry["vars"][k] = s
            per_block.append(entry)
        report[page] = per_block
    report["__assert_equal_held_to_1e-12__"] = len(loosened)
    with open(out_path, "wb") as f:
        pickle.dump(report, f)


# ---- parent: compare the two runs -----------------------------------------------------------------------------------------
def _same(a, b, path, out):
    import numpy as np

    if a is None or b is None or a.get("kind") != b.get("kind"):
        out.append(f"{path}: {None if a is None else a.get('kind')} vs {None if b is None else b.get('kind')}")
        return
    if a["kind"] == "DataArray":
        for key in ("dims", "name", "coords", "dtype"):
            if a[key] != b[key]:
                out.append(f"{path}: {key} {a[key]!r} vs {b[key]!r}")
                return
    if a["kind"] in ("DataArray", "ndarray"):
        x, y = a["values"], b["values"]
        if x.shape != y.shape:
            out.append(f"{path}: shape {x.shape} vs {y.shape}")
        elif not np.array_equal(x, y, equal_nan=x.dtype.kind == "f") and not np.allclose(x, y, rtol=1e-12, atol=1e-12, equal_nan=True):
            out.append(f"{path}: values differ, max |d| = {np.nanmax(np.abs(x.astype(float) - y.astype(float))):.3e}")
    elif a["kind"] == "Dataset":
        if sorted(a["vars"]) != sorted(b["vars"]):
            out.append(f"{path}: variables {sorted(a['vars'])} vs {sorted(b['vars'])}")
        else:
            for k in a["vars"]:
                _same(a["vars"][k], b["vars"][k], f"{path}.{k}", out)
    elif a["kind"] == "tuple":
        if len(a["items"]) != len(b["items"]):
            out.append(f"{path}: {len(a['items'])} vs {len(b['items'])} items")
        else:
            for i, (p, q) in enumerate(zip(a["items"], b["items"])):
                _same(p, q, f"{path}[{i}]", out)
    elif a["kind"] == "dict":
        if sorted(a["items"]) != sorted(b["items"]):
            out.append(f"{path}: keys differ")
        else:
            for k in a["items"]:
                _same(a["items"][k], b["items"][k], f"{path}[{k}]", out)


def run(backend="oracle-double", fused=False):
    if not os.path.isdir(os.path.join(REF, "docs")):
        raise FileNotFoundError(f"{REF}/docs: the reference is not on this box")
    results = {}
    for which in ("ref", "own"):
        fd, path = tempfile.mkstemp(suffix=".pkl", prefix=f"xgcm_docs_{which}_")
        os.close(fd)
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", which, "--backend", backend, "--out", path] + (["--fused"] if fused else [])
            proc = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
            if proc.returncode != 0:
                raise RuntimeError(f"{which} run failed:\n{proc.stderr[-3000:]}")
            with open(path, "rb") as f:
                results[which] = pickle.load(f)
        finally:
            os.unlink(path)
    summary = {"pages": {}, "differences": [], "assert_equal_held_to_1e-12": results["own"].get("__assert_equal_held_to_1e-12__", 0)}
    for page in PAGES:
        if backend == "host-abi" and page == "transform.ipynb":
            continue  # (`xg_transform_*` is not part of the host build of the ABI)
        ref, own = results["ref"][page], results["own"][page]
        n_vars = n_raised = 0
        for i, (r, o) in enumerate(zip(ref, own)):
            where = f"{page}[{i}]"
            if (r["raised"] is None) != (o["raised"] is None):
                summary["differences"].append(f"{where}: reference {r['raised']}, xgcm_amd {o['raised']}")
                continue
            if r["raised"] is not None:
                n_raised += 1
                if r["raised"][0] != o["raised"][0]:
                    summary["differences"].append(f"{where}: exception {r['raised']} vs {o['raised']}")
            names = sorted(set(r["vars"]) | set(o["vars"]))
            for k in names:
                found = []
                _same(r["vars"].get(k), o["vars"].get(k), f"{where}:{k}", found)
                summary["differences"] += found
                n_vars += 1
        summary["pages"][page] = {"blocks": len(ref), "blocks_raising_in_both": n_raised, "snapshots_compared": n_vars}
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--backend", default="oracle-double", choices=["oracle-double", "host-abi", "hip"])
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    if args.child:
        child(args.child, args.backend, args.fused, args.out)
        return
    summary = run(args.backend, args.fused)
    print(json.dumps(summary["pages"]))
    print("the guide's own assert_equal held to 1e-12 instead of to the bit:", summary["assert_equal_held_to_1e-12"])
    for d in summary["differences"][: (None if args.verbose else 30)]:
        print("DIFF", d[:300])
    print(len(summary["differences"]), "differences")
    sys.exit(1 if summary["differences"] else 0)


if __name__ == "__main__":
    main()
