#!/usr/bin/env python3
"""The reference's OWN test suite, unmodified, run against `xgcm_amd`.

TEST INFRASTRUCTURE -- build container only (reads /root/reference at run time; nothing of it is copied into the repo),
never shipped, never on the GPU box.

    python oracle/run_reference_suite.py [--backend host-abi|oracle-double|hip] [--report out.json] [pytest args]

How: a scratch directory under /tmp gets a package NAMED `xgcm` whose modules re-export `xgcm_amd`'s (`xgcm.grid` is
`xgcm_amd.grid`, ... -- the import lines of the reference's tests resolve to THIS package), its `test` sub-package is a
symlink to `/root/reference/xgcm/test` (the test files are read where they lie), a module NAMED `xarray` is
`oracle/xr_min.py` + `oracle/xr_suite.py` (xarray is not installable here: numpy-backed stand-in, see those files), `dask`
is an empty stub (tests that need chunked arrays are reported as `needs-dask`, the product refuses dask inputs by design,
DESIGN section 10), and `xgcm_amd.device` is one of the CPU test doubles (the host build of the C ABI by default) or, on a
GPU box that has the reference, the real library.  pytest then collects `xgcm/test/*.py` as the reference's CI does.

What a pass means: the reference's assertion, written by its authors against its own implementation, holds for
`xgcm_amd` under the stand-in's container semantics ("pinned modulo the stand-in", as DESIGN section 7 says for every
fixture made this way).  Every outcome is written to a JSON report; `tests/golden/reference_suite_report.json` is the
committed one and `tests/test_reference_suite_live.py` re-runs the suite wherever /root/reference exists and fails on any
test that passed in the committed report and no longer does.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("XGCM_REFERENCE", "/root/reference")

SHIM_MODULES = ["grid", "grid_ufunc", "padding", "axis", "metadata_parsers", "metrics", "transform", "gridops", "comodo", "sgrid"]

INIT = '''"""scratch shim: the name `xgcm` bound to xgcm_amd (written by oracle/run_reference_suite.py)"""
from xgcm_amd import *  # noqa
from xgcm_amd import Grid, Axis, as_grid_ufunc, apply_as_grid_ufunc  # noqa
import xgcm_amd as _pkg
__version__ = _pkg.__version__
'''

SHIM = '''"""scratch shim: xgcm.{name} is xgcm_amd.{source} (written by oracle/run_reference_suite.py)"""
import xgcm_amd.{source} as _m
globals().update({{k: v for k, v in vars(_m).items() if not (k.startswith("__") and k.endswith("__"))}})
if "{name}" == "padding":
    from oracle.refsuite_adapters import extras as _extras
    globals().update(_extras(_m))
'''

CONFTEST = '''"""scratch conftest (written by oracle/run_reference_suite.py): stand-in `xarray`, stub `dask`, a device for xgcm_amd"""
import importlib.util
import json
import os
import sys
import types

import pytest

ROOT = {root!r}
BACKEND = {backend!r}
REPORT = {report!r}
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load_as(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


try:  # the real package where it is installed (XGCM_USE_STANDIN=1 forces the stand-in); this image has none
    if os.environ.get("XGCM_USE_STANDIN") == "1":
        raise ImportError
    import xarray as xr
    import xarray.testing  # noqa: F401
except ImportError:
    xr = _load_as("xarray", os.path.join(ROOT, "oracle", "xr_min.py"))        # classes live in a module NAMED xarray
    suite = _load_as("xarray._suite", os.path.join(ROOT, "oracle", "xr_suite.py"))
    suite.extend(xr)

try:
    import dask.array  # noqa: F401  (installed: the chunked tests then meet the product's refusal of dask inputs)
    _have_dask = True
except ImportError:
    _have_dask = False
dask = types.ModuleType("dask")
dask_array = types.ModuleType("dask.array")
dask_array.Array = type("Array", (), {{}})
dask.array = dask_array
if not _have_dask:
    sys.modules["dask"] = dask
    sys.modules["dask.array"] = dask_array


def _needs_dask(*a, **k):
    pytest.skip("needs-dask: a dask scheduler / chunked arrays (dask is not installable here)")


import contextlib

dask.config = types.SimpleNamespace(set=lambda **k: contextlib.nullcontext())  # a scheduler choice alone needs no dask
dask_distributed = types.ModuleType("dask.distributed")
dask_distributed.Client = dask_distributed.LocalCluster = _needs_dask
dask.distributed = dask_distributed
sys.modules["dask.distributed"] = dask_distributed
dask_array.map_overlap = dask_array.from_array = dask_array.ones = dask_array.zeros = _needs_dask
# test_transform.py skips itself unless `import numba` works; xgcm_amd.transform needs no JIT (its kernels are HIP), so an
# empty module of that name is all the test module's guard asks for
sys.modules["numba"] = types.ModuleType("numba")


class _MP:
    """the two calls the doubles' install() makes on a pytest monkeypatch, for the whole session"""

    def setattr(self, target, name=None, value=None, raising=True):
        if value is None and isinstance(target, str):
            mod, _, attr = target.rpartition(".")
            import importlib

            setattr(importlib.import_module(mod), attr, name)
        else:
            setattr(target, name, value)

    def setitem(self, mapping, key, value):
        mapping[key] = value


if BACKEND == "host-abi":
    import host_abi_device

    host_abi_device.install(_MP())
elif BACKEND == "oracle-double":
    from oracle import fake_device

    fake_device.install(_MP())
elif BACKEND != "hip":
    raise RuntimeError(BACKEND)

if {fused!r}:
    # every Grid of the suite built with fuse=True: diff / interp / min / max / derivative return deferred results
    # (xgcm_amd.lazy) and the suite's assertions force them -- a differential test of the deferred mode against the
    # reference's expectations.  `xarray.testing.*` is where a deferred result meets an xarray object: there (only)
    # `.to_xarray()` is applied, the documented hand-over (DESIGN 1a)
    import xgcm_amd.grid as _g
    import xgcm_amd.lazy as _lz

    _init = _g.Grid.__init__

    def _fused_init(self, *a, **k):
        k.setdefault("fuse", True)
        _init(self, *a, **k)

    _g.Grid.__init__ = _fused_init
    for _name in ("assert_allclose", "assert_equal", "assert_identical"):
        def _wrap(f):
            def g(a, b, *rest, **kw):
                a, b = (v.to_xarray() if isinstance(v, _lz.LazyArray) else v for v in (a, b))
                return f(a, b, *rest, **kw)
            return g
        setattr(xr.testing, _name, _wrap(getattr(xr.testing, _name)))

_results = {{}}


def pytest_runtest_logreport(report):
    key = report.nodeid
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        entry = {{"outcome": report.outcome}}
        if report.outcome != "passed":
            text = str(report.longrepr)
            lines = [ln for ln in text.strip().splitlines() if ln.startswith("E  ")]
            entry["why"] = (lines[0][1:].strip() if lines else (text.strip().splitlines()[-1] if text.strip() else ""))[:300]
            if report.outcome == "skipped" and isinstance(report.longrepr, tuple):
                entry["why"] = str(report.longrepr[2])[:300]
        if hasattr(report, "wasxfail"):
            entry["outcome"] = "xfailed" if report.outcome == "skipped" else "xpassed"
        _results[key] = entry


def pytest_collectreport(report):
    if report.failed:
        _results[report.nodeid] = {{"outcome": "collect-error", "why": str(report.longrepr).strip().splitlines()[-1][:300]}}


def pytest_sessionfinish(session, exitstatus):
    with open(REPORT, "w") as f:
        json.dump(_results, f, indent=0, sort_keys=True)
'''


def build_scratch(backend, report, fused=False):
    scratch = tempfile.mkdtemp(prefix="xgcm_refsuite_")
    pkg = os.path.join(scratch, "xgcm")
    os.makedirs(pkg)
    with open(os.path.join(pkg, "__init__.py"), "w") as f:
        f.write(INIT)
    for name in SHIM_MODULES:
        source = {"metadata_parsers": "metadata", "comodo": "metadata", "sgrid": "metadata"}.get(name, name)
        with open(os.path.join(pkg, name + ".py"), "w") as f:
            f.write(SHIM.format(name=name, source=source))
    os.symlink(os.path.join(REF, "xgcm", "test"), os.path.join(pkg, "test"))
    with open(os.path.join(scratch, "conftest.py"), "w") as f:
        f.write(CONFTEST.format(root=ROOT, backend=backend, report=report, fused=bool(fused)))
    return scratch


def run(backend="host-abi", report=None, extra=(), quiet=True, fused=False):
    if not os.path.isdir(os.path.join(REF, "xgcm", "test")):
        raise FileNotFoundError(f"{REF}/xgcm/test: the reference is not on this box")
    own_report = report is None
    if own_report:
        fd, report = tempfile.mkstemp(suffix=".json", prefix="xgcm_refsuite_")
        os.close(fd)
    scratch = build_scratch(backend, report, fused)
    try:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=scratch + os.pathsep + ROOT)
        cmd = [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "--rootdir", scratch, "-c", os.devnull,
               "-o", "python_files=test_*.py", "-q", "--no-header", "-W", "ignore",
               *(extra or [os.path.join(scratch, "xgcm", "test")])]
        proc = subprocess.run(cmd, cwd=scratch, env=env, capture_output=quiet, text=True)
        with open(report) as f:
            results = json.load(f)
        return results, proc
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
        if own_report:
            os.unlink(report)


def summarize(results):
    counts = {}
    for v in results.values():
        counts[v["outcome"]] = counts.get(v["outcome"], 0) + 1
    return counts


def by_function(results):
    """{"file::[Class::]function": {outcome: count}} -- the committed form (ids of 4000 parametrised cases would be noise)"""
    table = {}
    for nodeid, v in results.items():
        key = nodeid.split("[", 1)[0]
        row = table.setdefault(key, {})
        row[v["outcome"]] = row.get(v["outcome"], 0) + 1
    return dict(sorted(table.items()))


HOST_BUILD_GAP = "not part of the host build of the ABI"


def committed_report():
    """Both CPU backends, in the form `tests/golden/reference_suite_report.json` holds"""
    out = {"what": "outcomes of the reference's own test suite (xgcm/test/*.py, unmodified, read in place) run against xgcm_amd "
                   "by oracle/run_reference_suite.py; pinned modulo the xarray stand-in (oracle/xr_min.py + xr_suite.py)",
           "backends": {}}
    for backend in ("oracle-double", "oracle-double+fused", "host-abi"):
        results, _ = run(backend.split("+")[0], fused=backend.endswith("+fused"))
        failed = {k: v.get("why", "") for k, v in results.items() if v["outcome"] in ("failed", "collect-error")}
        skipped = {}
        for v in results.values():
            if v["outcome"] == "skipped":
                why = v.get("why", "").replace("Skipped: ", "")[:60]
                skipped[why] = skipped.get(why, 0) + 1
        out["backends"][backend] = {"summary": summarize(results), "skip_reasons": skipped,
                                    "failed": dict(sorted(failed.items())) if backend != "host-abi" else
                                    {"count": len(failed), "all_outside_the_host_build": all(HOST_BUILD_GAP in w or "XgcmHipError" in w for w in failed.values())},
                                    "functions": by_function(results)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="host-abi", choices=["host-abi", "oracle-double", "hip"])
    ap.add_argument("--report", default=None)
    ap.add_argument("--show", action="store_true", help="pytest's own output")
    ap.add_argument("--fused", action="store_true", help="every Grid with fuse=True (deferred results, xgcm_amd.lazy)")
    ap.add_argument("--write-report", action="store_true", help="run both CPU backends, rewrite tests/golden/reference_suite_report.json")
    args, extra = ap.parse_known_args()
    if args.write_report:
        rep = committed_report()
        with open(os.path.join(ROOT, "tests", "golden", "reference_suite_report.json"), "w") as f:
            json.dump(rep, f, indent=1, sort_keys=True)
        print(json.dumps({b: v["summary"] for b, v in rep["backends"].items()}))
        return
    results, proc = run(args.backend, args.report, extra, quiet=not args.show, fused=args.fused)
    if not args.show:
        print(proc.stdout[-3000:])
        print(proc.stderr[-2000:], file=sys.stderr)
    print(json.dumps(summarize(results)))


if __name__ == "__main__":
    main()
