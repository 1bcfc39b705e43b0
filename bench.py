#!/usr/bin/env python3
"""Headline benchmark: Grid.interp + Grid.diff on a 3600 x 2400 x 75 float64 MITgcm-like C-grid
(BASELINE.json configs[1]), periodic X / extend Y, through the public `xgcm_amd.Grid` API.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one resident record (field T(Z,YC,XC), 648e6 cells):
interp(T,'X'), diff(T,'X') [contiguous axis, periodic] and interp(T,'Y'), diff(T,'Y') [strided
axis, extend] = 4 kernel launches, 4 x 648e6 output cells.  Inputs are generated in HBM before
the timed region (synthetic, bit-identical to the oracle's generator); outputs stay in HBM.
Multi-GPU: every rank owns its own record(s) (the path shards over the outer record axis with no
data-path collective); RCCL is used for the barriers and the max-time
reduction only => weak scaling, value = cells of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the numpy oracle = the reference's eager call sequence, timed on a
bounded sample on this box's host cores; N=1 only).

AFTER the headline's timed region (value / ms_per_step are not touched by it) the same process puts
one-rank shares of BASELINE configs[2..4] under the same clock and adds them to the line as `configs`
(`--no-configs` skips it): config 3 `derivative` X / Y / Z with random 2-D / 1-D metrics + `integrate` Z,
config 4 `cumsum` Z center->left / center->outer over a resident batch of 3 records, config 5 the fused
vorticity and the chain AS WRITTEN under `grid.fused()` on 4320 x 4320 x 90 -- each with HIP-event ms,
fraction of 8 TB/s on SURVEY section 8(d)'s bytes, and a bit check of a slab against the oracle (made in
the cpu_baseline leg, after every timed span).  `box_probe` = three launches of cumsum Z on ONE record:
the figure that tells a fast box (1.65 - 1.70 ms) from a slow one (2.0), DESIGN section 8.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

NZ, NY, NX = 75, 2400, 3600
CONFIG5_SHAPE = (90, 4320, 4320)
BASELINE_SHAPES = ((NZ, NY, NX), CONFIG5_SHAPE)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 16.0  # 1 f64 read + 1 f64 write per output cell
OPS = [("interp", "X"), ("diff", "X"), ("interp", "Y"), ("diff", "Y")]


class Mark:
    """a point on the launch stream: a HIP event (torch's current stream = the stream the kernels go to).  Without a GPU the
    device layer raises long before a Mark is made -- except in the CPU dry run of tests/test_bench_dryrun.py, where the TEST
    harness has swapped the device layer's memory for the host build of the C ABI to drive this file's launcher / rank / JSON
    path with 8 gloo ranks; the marks are then host clock readings (no number of such a run is a measurement)."""

    def __init__(self):
        if torch.cuda.is_available():
            self.ev = torch.cuda.Event(enable_timing=True)
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def ms_until(self, later: "Mark") -> float:
        if torch.cuda.is_available():
            return self.ev.elapsed_time(later.ev)
        return (later.t - self.t) * 1e3


def _empty_cache():
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def kernel_of_axis():
    """the kernel that serves each axis of this workload (profiles/*_rocprof_summary_bench.txt lists them by these names)"""
    from xgcm_amd import _hip

    return {"X": "k_stencil_contig", "Y": "k_stencil_strided_ys" if _hip.get_tunable("seg_ys") else "k_stencil_strided_seg"}


def build_grid(nz, device_field):
    from xgcm_amd import DataArray, Dataset, Grid

    coords = {"XC": ("XC", np.arange(NX) + 0.5), "XG": ("XG", np.arange(NX) * 1.0),
              "YC": ("YC", np.arange(NY) + 0.5), "YG": ("YG", np.arange(NY) * 1.0),
              "Z": ("Z", np.arange(nz) * 1.0)}
    ds = Dataset(coords=coords)
    grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"},
                            "Z": {"center": "Z"}},
                padding={"X": "periodic", "Y": "extend"}, autoparse_metadata=False)
    T = DataArray(device_field, ("Z", "YC", "XC"), name="T")
    return grid, T


def _cpu_pass(a):
    from oracle import refimpl as R

    R.stencil1d("interp", a, 2, 1, 0, "periodic")
    R.stencil1d("diff", a, 2, 1, 0, "periodic")
    R.stencil1d("interp", a, 1, 1, 0, "extend")
    R.stencil1d("diff", a, 1, 1, 0, "extend")
    return 4 * a.size


def spot_check(spot):
    """Part of the cpu_baseline leg: level 0 of the HIP results (diff X periodic, interp Y extend) against
    the oracle, bit for bit."""
    from oracle import refimpl as R

    a0 = R.synthetic_field((1, NY, NX), 2)
    return bool(np.array_equal(spot[0], R.stencil1d("diff", a0, 2, 1, 0, "periodic"))
                and np.array_equal(spot[1], R.stencil1d("interp", a0, 1, 1, 0, "extend")))


def cpu_baseline(levels=8, budget_s=20.0):
    """The reference's eager numpy sequence (oracle/refimpl.py) on a `levels`-deep slab of the same
    workload.  `value`: single thread = the reference's own eager execution model (numpy, no dask).
    `threaded`: the same sequence on one level per task in a thread pool (numpy releases the GIL) =
    stand-in for the reference's dask threaded scheduler over a field chunked along Z; `dask_threaded`: that scheduler
    itself, where a dask can be found on the box."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import refimpl as R

    a = R.synthetic_field((levels, NY, NX), 2)
    cells = 0
    t0 = time.perf_counter()
    passes = 0
    while True:
        cells += _cpu_pass(a)
        passes += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.4 or passes >= 8:
            break
    out = {"value": round(cells / el / 1e9, 4), "unit": "Gcell/s", "cores": 1, "kind": "port",
           # what the port restates (the reference is Python over xarray: not importable on the GPU box): per axis
           # `DataArray.pad` = numpy.pad copy, then the sliced two-point body on the padded array
           "restates": "oracle/refimpl.py::stencil1d = xgcm/grid.py:728-836 (_1d_grid_ufunc_dispatch, one grid ufunc per axis) -> "
                       "xgcm/grid_ufunc.py:885-904 (pad, then apply) -> xgcm/padding.py:575-616 (_pad_basic: numpy.pad wrap / edge) -> "
                       "xgcm/gridops.py:23-24, 76-77 (diff_forward, interp_forward); pinned bit for bit by tests/golden/gridops_vectors.npz",
           "sample": f"{passes} pass(es) of the 4 ops on a {levels}x{NY}x{NX} f64 slab (numpy pad copy + sliced op, "
                     f"single thread = the reference's eager path), {el:.1f} s; host has {os.cpu_count()} cores"}
    # every host core (the dask threaded scheduler's default), bounded only by host memory: a task holds its
    # level, the padded copy, the sliced views' result and the output (~6 levels of 69 MB)
    nthreads = os.cpu_count() or 1
    try:
        avail = next(int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable"))
        nthreads = max(1, min(nthreads, int(0.5 * avail // (6 * NY * NX * 8))))
    except Exception:
        pass
    big = R.synthetic_field((nthreads, NY, NX), 2)
    chunks = [big[i:i + 1] for i in range(nthreads)]
    t0 = time.perf_counter()
    tcells = 0
    rounds = 0
    with ThreadPoolExecutor(nthreads) as pool:
        while True:
            tcells += sum(pool.map(_cpu_pass, chunks))
            rounds += 1
            el = time.perf_counter() - t0
            if el > budget_s * 0.4 or rounds >= 4:
                break
    out["threaded"] = {"value": round(tcells / el / 1e9, 4), "unit": "Gcell/s", "cores": nthreads,
                       "what": "NOT dask: the same numpy sequence, one level per task in a thread pool (numpy releases the GIL) -- a memory-bound "
                               "stand-in for the reference's dask threaded scheduler over a field chunked along Z (xgcm/grid.py:786-789)",
                       "host_cores": os.cpu_count(), "numpy": np.__version__,
                       "sample": f"{rounds} round(s), one 1x{NY}x{NX} level per task on {nthreads} threads "
                                 f"(all {os.cpu_count()} host cores unless host memory bounds it), {el:.1f} s"}
    # ... and the reference's parallel execution model itself where a dask can be found (the image's Anaconda tree holds a
    # pure-Python one: oracle/real_dask.py): a field chunked one level per chunk, the numpy sequence mapped over its blocks --
    # what `apply_ufunc(dask="parallelized")` builds (xgcm/grid.py:786-818) -- computed by the threaded scheduler
    try:
        from oracle.real_dask import dask_array

        dsa = dask_array()
    except Exception:  # noqa: BLE001 -- the leg is optional
        dsa = None
    if dsa is not None:
        import dask

        field = dsa.from_array(big, chunks=(1, NY, NX))
        graph = field.map_blocks(lambda b: np.full((1, 1, 1), float(_cpu_pass(b))), chunks=((1,) * nthreads, (1,), (1,)), dtype="f8")
        t0 = time.perf_counter()
        dcells, rounds = 0, 0
        while True:
            dcells += int(graph.sum().compute(scheduler="threads", num_workers=nthreads))
            rounds += 1
            el = time.perf_counter() - t0
            if el > budget_s * 0.4 or rounds >= 4:
                break
        out["dask_threaded"] = {"value": round(dcells / el / 1e9, 4), "unit": "Gcell/s", "cores": nthreads, "dask": dask.__version__,
                                "what": "REAL dask: the field as a dask array of one level per chunk, the same numpy sequence mapped over its "
                                        "blocks (what apply_ufunc(dask='parallelized') builds, xgcm/grid.py:786-818), threaded scheduler",
                                "sample": f"{rounds} round(s) of {nthreads} chunks of 1x{NY}x{NX} on {nthreads} threads, {el:.1f} s"}
    return out


def _kernel_base(name: str) -> str:
    """`void (anonymous namespace)::k_stencil_strided_ys<0, 1>(Args) [clone .kd]` -> `k_stencil_strided_ys` (every kernel of the
    library is named k_...; the name up to its template list, so that `..._ys` does not also match `..._ysm<...>`)"""
    import re

    m = re.search(r"\b(k_[A-Za-z0-9_]+)", name)
    return m.group(1) if m else ""


def pmc_passes(dominant: str, levels: int, alg_bytes: float, timeout_s: int = 120):
    """HBM bytes per launch of the dominant kernel measured NOW: two child runs of this file under `rocprofv3 --pmc`, one
    counter per pass (FETCH_SIZE, WRITE_SIZE) with `--kernel-trace` only -- never combined with another trace domain --
    and the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts KiB and reports half of a wide streaming read:
    bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024).  Returns (bytes or None, how / why not, seconds spent)."""
    import glob
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile

    t0 = time.perf_counter()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", 0.0
    tmp = tempfile.mkdtemp(prefix="xg_pmc_", dir="/tmp")
    env = dict(os.environ, XG_BENCH_PMC="0", XG_BENCH_PRIME="1", TMPDIR="/tmp")
    kib = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "1", "--levels", str(levels), "--no-cpu-baseline", "--no-pmc", "--no-configs"]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                return None, f"the {counter} pass did not finish in {timeout_s} s", time.perf_counter() - t0
            dbs = sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True))
            if proc.returncode != 0 or not dbs:
                return None, f"the {counter} pass failed (rc {proc.returncode})", time.perf_counter() - t0
            rows = sqlite3.connect(dbs[0]).execute(
                "select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
            # full-size launches of the dominant kernel (the one-level spot check launches the same kernels on 1 / 75 of the data)
            # (the name up to its template list: `k_stencil_strided_ys` must not also match `k_stencil_strided_ysm<...>`)
            vals = [v for name, v in rows if _kernel_base(name) == dominant and v * 1024 > 0.2 * alg_bytes]
            if not vals:
                return None, f"no {dominant} dispatch in the {counter} pass", time.perf_counter() - t0
            kib[counter] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return kib["FETCH_SIZE"] * 1024 * 2 + kib["WRITE_SIZE"] * 1024, "measured", time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------------------------
# `configs`: BASELINE configs[2..4] under the driver's clock (VERDICT r05 "next round" 1).  reference: xgcm/grid.py:1534-1605
# (derivative / integrate), :1183-1418 (cumsum), docs/ufunc_examples.md:96-311 (vorticity)
# ---------------------------------------------------------------------------------------------------------------------
def _timed(fn, reps, sync, warm_s=0.15):
    """HIP events on the launch stream around `reps` back-to-back calls, after at least `warm_s` of untimed calls
    (allocator steady state, clocks); returns (median ms, mean ms, last result)"""
    t0 = time.perf_counter()
    out = fn()
    sync()
    calls = 1
    while time.perf_counter() - t0 < warm_s or calls < 2:
        out = fn()
        sync()
        calls += 1
    ev = [Mark()]
    for i in range(reps):
        out = fn()
        ev.append(Mark())
    sync()
    ts = sorted(ev[i].ms_until(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2], float(np.mean(ts)), out


def _entry(name, kernel_bytes_per_cell, cells, med_ms, mean_ms, **extra):
    gbs = cells * kernel_bytes_per_cell / (med_ms * 1e-3) / 1e9
    e = {"op": name, "ms": round(med_ms, 4), "mean_ms": round(mean_ms, 4), "cells": int(cells),
         "bytes_per_cell": round(kernel_bytes_per_cell, 4), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBPS, 4)}
    e.update(extra)
    return e


def run_configs(ranks, field, reps, records4):
    """One rank's share of configs[2..4], every rank of the job at once (barrier before each config; no data-path
    collective).  Returns (block, slabs): `block` goes into the JSON line, `slabs` are small host copies of inputs and
    of the HIP results that `check_configs` hands to the oracle AFTER all timed spans."""
    from tools.bench_configs import config5_grid, mitgcm_grid
    from xgcm_amd import DataArray
    from xgcm_amd import device as D
    from xgcm_amd import sharding as S

    world, rank = ranks.world, ranks.rank
    nz, ny, nx = NZ, NY, NX
    cells = nz * ny * nx
    block, slabs = {}, {}
    t_all = time.perf_counter()

    def gather(entries):
        """per-op figures of every rank -> rank 0's entry carries the slowest rank's time and the job's aggregate"""
        if world == 1:
            return entries
        per_rank = ranks.gather_objects([(e["ms"], e["cells"]) for e in entries])
        if rank == 0:
            for i, e in enumerate(entries):
                ms = [r[i][0] for r in per_rank]
                tot = sum(r[i][1] for r in per_rank)
                e["per_rank_ms"] = ms
                e["job_gcell_s"] = float(f"{tot / (max(ms) * 1e-3) / 1e9:.6g}")  # (significant digits: a CPU dry run's rate is tiny, not 0)
        return entries

    # ---- config 3: derivative X / Y / Z (random metrics dxC(YC,XG), dyC(YG,XC), drC(Zl)) and integrate Z (drF(Z)) ----
    ranks.barrier()
    grid = mitgcm_grid(nz, ny, nx)
    T = DataArray(field, ("Z", "YC", "XC"), name="T")
    c3 = []
    for name, fn, bpc in (("derivative(T,'X') / dxC(YC,XG), periodic", lambda: grid.derivative(T, "X"), 16 + 8 / nz),
                          ("derivative(T,'Y') / dyC(YG,XC), extend", lambda: grid.derivative(T, "Y"), 16 + 8 / nz),
                          ("derivative(T,'Z') / drC(Zl), fill", lambda: grid.derivative(T, "Z"), 16.0),
                          ("integrate(T,'Z') * drF(Z)", lambda: grid.integrate(T, "Z"), 8 + 8 / nz)):
        med, mean, out = _timed(fn, reps, sync)
        c3.append(_entry(name, bpc, cells, med, mean))
        key = name.split(" ")[0]
        if rank == 0:
            if "'Z'" in name:   # all levels of two rows
                slabs[key] = (D.tohost(field[:, :2].contiguous()), D.tohost(out.data[..., :2, :].contiguous()))
            else:               # level 0
                slabs[key] = (D.tohost(field[:1].contiguous()), D.tohost(out.data[:1].contiguous()))
        del out
    if rank == 0:
        slabs["metrics"] = {k: D.tohost(grid._ds[k].data) for k in ("dxC", "dyC", "drF", "drC")}
    block["config3"] = {"workload": f"configs[2]: Grid.derivative X/Y/Z + Grid.integrate Z on {nx}x{ny}x{nz} f64 with random dx/dy/dz metrics, one record per GPU",
                        "ops": gather(c3)}

    # ---- box probe: three launches of cumsum Z on ONE record (1.65 - 1.70 ms fast boxes, 2.0 slow ones) ----
    grid.cumsum(T, "Z")
    sync()
    ev = [Mark()]
    for i in range(3):
        grid.cumsum(T, "Z")
        ev.append(Mark())
    sync()
    probe = [round(ev[i].ms_until(ev[i + 1]), 4) for i in range(3)]
    block["box_probe"] = {"op": f"cumsum(T,'Z') center->left on one {nx}x{ny}x{nz} f64 record, 3 launches", "ms": probe,
                          "frac": round(cells * 16 / (min(probe) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    if world > 1:
        allp = ranks.gather_objects(probe)
        if rank == 0:
            block["box_probe"]["per_rank_ms"] = allp
    del T
    _empty_cache()

    # ---- config 4: cumsum Z over a resident batch of this rank's records (360 records over the record axis) ----
    lo, hi = S.shard_bounds(360, world, rank)
    nrec = max(1, min(records4, hi - lo))
    T4 = DataArray(D.synthetic((nrec, nz, ny, nx), 4, offset=lo * cells), ("time", "Z", "YC", "XC"))
    c4 = []
    for to in ("left", "outer"):
        kw = {} if to == "left" else {"to": to}
        ranks.barrier()
        med, mean, out = _timed(lambda: grid.cumsum(T4, "Z", **kw), reps, sync, warm_s=0.3)
        c4.append(_entry(f"cumsum(T,'Z') center->{to}, fill, {nrec} resident records of {nx}x{ny}x{nz} f64", 16.0, nrec * cells, med, mean,
                         ms_per_record=round(med / nrec, 4)))
        if rank == 0:
            slabs["cumsum_" + to] = (D.tohost(T4.data[nrec - 1, :, :2].contiguous()), D.tohost(out.data[nrec - 1, :, :2].contiguous()))
        del out
        _empty_cache()
    block["config4"] = {"workload": f"configs[3]: Grid.cumsum along Z (75 levels) on 3600x2400x75x360 sharded over t: this job's ranks each scan a resident "
                                    f"batch of {nrec} of their {hi - lo} records", "ops": gather(c4)}
    del T4, grid
    _empty_cache()

    # ---- config 5: vorticity on 4320 x 4320 x 90 split along Z over the ranks, `fill` ----
    nz5, n5y, n5 = CONFIG5_SHAPE
    l5, h5 = S.shard_bounds(nz5, world, rank)
    nl = h5 - l5
    g5 = config5_grid(n5y, n5)
    plane = n5y * n5
    c5 = []
    if nl:
        U = DataArray(D.synthetic((nl, n5y, n5), 51, offset=l5 * plane), ("Z", "YC", "XG"))
        V = DataArray(D.synthetic((nl, n5y, n5), 52, offset=l5 * plane), ("Z", "YG", "XC"))
        area = g5._ds["rAz"].reset_coords(drop=True)

        def as_written():
            with g5.fused():
                zeta = (g5.diff(V, "X") - g5.diff(U, "Y")) / area
            zeta.data  # the use
            return zeta

        def chain():
            return (g5.diff(V, "X") - g5.diff(U, "Y")) / area

        bpc5 = 24 + 8 / nl
        ranks.barrier()
        for name, fn, n in (("vorticity fused: grid.vorticity(U,V) = (diff(v,X)-diff(u,Y))/rAz, one launch", lambda: g5.vorticity(U, V), reps),
                            ("vorticity chain AS WRITTEN under grid.fused(): (grid.diff(V,'X') - grid.diff(U,'Y')) / rAz", as_written, reps),
                            ("vorticity chain eager (4 launches, the reference's order), fused-equivalent bytes", chain, max(3, reps // 2))):
            med, mean, out = _timed(fn, n, sync, warm_s=0.3)
            c5.append(_entry(name, bpc5, nl * plane, med, mean))
            if rank == 0:
                slabs.setdefault("vorticity", []).append(D.tohost(out.data[:1].contiguous()))
            del out
        if rank == 0:
            slabs["vorticity_in"] = (D.tohost(U.data[:1].contiguous()), D.tohost(V.data[:1].contiguous()), D.tohost(area.data))
        del U, V
    else:
        ranks.barrier()
    if world > 1 and not nl:  # more ranks than levels: still part of the gather
        c5 = [_entry("(no levels on this rank)", 24.0, 0, 1.0, 1.0) for _ in range(3)]
    levels = [S.shard_bounds(nz5, world, r)[1] - S.shard_bounds(nz5, world, r)[0] for r in range(world)]
    block["config5"] = {"workload": f"configs[4]: chained vorticity (diff(v,'X')-diff(u,'Y'))/area on {n5}x{n5y}x{nz5} f64, fill, split along Z over the ranks "
                                    f"(levels per rank {levels})", "ops": gather(c5)}
    _empty_cache()
    ranks.barrier()
    block["seconds"] = round(time.perf_counter() - t_all, 1)
    return block, slabs


def check_configs(block, slabs):
    """Part of the cpu_baseline leg: the slabs the timed spans kept (host copies of a few rows of inputs and HIP results)
    against the oracle, bit for bit."""
    from oracle import refimpl as R

    m = slabs["metrics"]
    eq = lambda a, b: bool(a.shape == b.shape and np.array_equal(a, b, equal_nan=True))  # noqa: E731
    a, got = slabs["derivative(T,'X')"]
    ok3 = [eq(got, R.derivative(a, 2, 1, 0, "periodic", 0.0, m["dxC"][None]))]
    a, got = slabs["derivative(T,'Y')"]
    ok3.append(eq(got, R.derivative(a, 1, 1, 0, "extend", 0.0, m["dyC"][None])))
    a, got = slabs["derivative(T,'Z')"]
    ok3.append(eq(got, R.derivative(a, 0, 1, 0, "fill", 0.0, m["drC"][:, None, None])))
    a, got = slabs["integrate(T,'Z')"]
    ok3.append(eq(got, R.integrate(a, 0, m["drF"][:, None, None])))
    for e, ok in zip(block["config3"]["ops"], ok3):
        e["bit_exact_vs_oracle"] = ok
    for e, to in zip(block["config4"]["ops"], ("left", "outer")):
        a, got = slabs["cumsum_" + to]
        e["bit_exact_vs_oracle"] = eq(got, R.grid_cumsum(a, 0, "center", to, "fill", 0.0))
    if "vorticity_in" in slabs:
        u, v, area = slabs["vorticity_in"]
        want = R.vorticity(u, v, area[None], "fill", "fill")
        for e, got in zip(block["config5"]["ops"], slabs["vorticity"]):
            e["bit_exact_vs_oracle"] = eq(got, want)
    block["checked"] = ("level 0 (X, Y ops, vorticity) or all levels of rows 0-1 (Z ops; last record of the batch) of every HIP result "
                        "against oracle/refimpl.py on the same slab of the input, numpy.array_equal")



def main():
    global NZ, NY, NX, CONFIG5_SHAPE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--levels", type=int, default=NZ, help="Z levels (default = the full 75; smaller only for debugging)")
    ap.add_argument("--shape", default="", help="Z,Y,X of the record and Z,Y,X of config 5, ';'-separated -- plumbing checks only (the CPU dry run of "
                    "tests/test_bench_dryrun.py): a line from another shape than BASELINE's says so in `config.workload` and is no measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (configs[2..4] after the headline's timed region)")
    ap.add_argument("--config4-records", type=int, default=3, help="`configs`: records of 3600x2400x75 in the resident batch of config 4 (per rank).  "
                    "3 records = 15.6 GB in, 15.6 - 15.8 GB out: the largest batch whose RESULT still comes from the library's scattered-buffer pool "
                    "(device.SCATTER_MAX_BYTES = 16 GiB); larger results are plain allocations and meet the placement lottery "
                    "(6 records: 0.67 - 0.79 from process to process, 3 records: 0.77 - 0.79; profiles/r06w_config4_batch_sizes.log)")
    ap.add_argument("--config-reps", type=int, default=7, help="`configs`: timed launches per operator")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child passes (N = 1 only; "
                    "also XG_BENCH_PMC=0): the committed figure of profiles/pmc_traffic.json is reported instead")
    args = ap.parse_args()
    if args.shape:
        rec, _, c5 = args.shape.partition(";")
        NZ, NY, NX = (int(v) for v in rec.split(","))
        if c5:
            CONFIG5_SHAPE = tuple(int(v) for v in c5.split(","))
        if args.levels == 75:
            args.levels = NZ
    toy = ((NZ, NY, NX), CONFIG5_SHAPE) != BASELINE_SHAPES

    from xgcm_amd import sharding as S

    # `--gpus N` means N ranks whoever started us: under torchrun we ARE one of them (WORLD_SIZE must agree),
    # started by hand we re-execute under torch.distributed.run; fewer visible GPUs than ranks is an error
    S.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    # RCCL ("nccl") wherever there is a GPU; XG_DIST_BACKEND=gloo only in the CPU dry run (tests/test_bench_dryrun.py)
    ranks = S.init_ranks(args.gpus, backend=os.environ.get("XG_DIST_BACKEND") or "nccl")
    world, rank, local_rank, dist = ranks.world, ranks.rank, ranks.local_rank, ranks.dist

    import __graft_entry__ as entry

    if entry._stale():  # clean checkout: the .so is a git-ignored build artefact
        if rank == 0:
            entry.build()
        ranks.barrier()
    from xgcm_amd import device as D

    nz = args.levels
    cells_per_op = nz * NY * NX
    # each rank generates ITS record (seed 2, disjoint index range) directly in HBM
    field = D.synthetic((nz, NY, NX), 2, offset=rank * cells_per_op)
    grid, T = build_grid(nz, field)

    def step(marks=None):
        for i, (fn, ax) in enumerate(OPS):
            getattr(grid, fn)(T, ax)
            if marks is not None:
                marks.append(Mark())

    barrier = ranks.barrier  # synchronize, barrier over the process group (RCCL), synchronize

    # one level through the HIP path, kept for the parity spot-check that the cpu_baseline leg makes
    # against the oracle after the timed region (the oracle is touched nowhere else in this file)
    spot = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        g1, T1 = build_grid(1, field[:1].contiguous())
        spot = (g1.diff(T1, "X").values, g1.interp(T1, "Y").values)

    # untimed priming (allocator pool, code-object load, clock ramp) so that small --warmup values do
    # not leak one-off start-up stalls into the timed region; then the W warmup steps as contracted
    prime = int(os.environ.get("XG_BENCH_PRIME", "5"))
    for _ in range(prime):
        step()
    sync()
    import gc

    K = args.steps
    ev = [[] for _ in range(K)]
    # the collector runs BEFORE the warmup steps: a collection between warmup and the timed region left the GPU idle long
    # enough to drop its clocks, and the first timed step then ran 0.8 ms (12 %) slower than the others
    # (`device_ms_per_step.first` against `.median`; three runs on one box: slowest_step = 0 every time)
    gc.collect()
    gc.disable()  # an interpreter GC pause (~10 ms, seen once in r01d) is not part of the hot path
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k].append(Mark())
        step(ev[k])
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()

    local_cells = K * len(OPS) * cells_per_op
    per_rank_ms_per_step = [round(v / K * 1e3, 4) for v in ranks.gather_floats(elapsed)]
    # what a < 7x curve would have to be explained with: where each rank ran (GPU, NUMA node, CPUs it was bound to) and how
    # much of its wall time per step was NOT device time (Python dispatch of the 4 launches, barrier skew)
    local_dev_ms = sorted(ev[k][0].ms_until(ev[k][len(OPS)]) for k in range(K))
    placement = ranks.gather_objects(dict(ranks.placement, rank=rank,
                                          device_ms_per_step=round(local_dev_ms[len(local_dev_ms) // 2], 4),
                                          host_overhead_ms_per_step=round(elapsed / K * 1e3 - float(np.mean(local_dev_ms)), 4)))
    cells_per_s, elapsed = S.whole_job_throughput(local_cells, elapsed, dist, ranks.scalar_device)
    n_ranks = dist.get_world_size() if dist is not None else 1  # the rank count RCCL itself reports

    # per-launch durations from the HIP events recorded on the launch stream inside the timed region
    per_op_ms = [float(np.mean([ev[k][i].ms_until(ev[k][i + 1]) for k in range(K)])) for i in range(len(OPS))]
    step_ms_in_order = [ev[k][0].ms_until(ev[k][len(OPS)]) for k in range(K)]
    step_ms = sorted(step_ms_in_order)  # device time per step
    by_kernel, of_axis = {}, kernel_of_axis()
    for (fn, ax), ms in zip(OPS, per_op_ms):
        by_kernel.setdefault(of_axis[ax], []).append(ms)
    dominant = max(by_kernel, key=lambda k: sum(by_kernel[k]))
    dom_ms = float(np.mean(by_kernel[dominant]))
    alg_bytes = cells_per_op * BYTES_PER_CELL
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9

    cfg_block = cfg_slabs = None
    if not args.no_configs and nz == NZ:
        cfg_block, cfg_slabs = run_configs(ranks, field, args.config_reps, args.config4_records)

    if rank == 0:
        total_cells = cells_per_s * elapsed
        # HBM bytes per launch of the dominant kernel: NOT measured in this run (counter passes serialise the kernels and
        # need rocprofv3 around the process) but in separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes of this
        # same command, whose summary is committed next to the number; `traffic_source` names it
        traffic, traffic_source, measured_now, pmc_s = None, None, False, 0.0
        under_profiler = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
        if world == 1 and dist is None and not args.no_pmc and os.environ.get("XG_BENCH_PMC", "1") != "0" and not under_profiler:
            # N = 1: measured in THIS run, after the timed region, by two short child passes of this command
            traffic, how, pmc_s = pmc_passes(dominant, nz, alg_bytes)
            if traffic is not None:
                measured_now = True
                traffic_source = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command after the timed region "
                                  f"({pmc_s:.0f} s; FETCH_SIZE doubled per MI355X_MICROARCH.md)")
            else:
                traffic_source = f"live PMC passes unavailable ({how}); "
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(pmc) and not toy:  # (the committed passes are of BASELINE's shape)
            try:
                table = json.load(open(pmc))
                traffic = table.get(dominant)
                if traffic is not None:
                    traffic_source = (traffic_source or "") + "offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " + \
                        str(table.get("_source", "profiles/pmc_traffic.json"))
            except Exception:
                traffic = None
        line = {
            "metric": f"stencil-cells/s, Grid.interp+Grid.diff (X periodic, Y extend) on {NX}x{NY}x{nz} f64",
            "value": round(total_cells / elapsed / 1e9, 3),
            "unit": "Gcell/s",
            "n_gpus": n_ranks,
            "steps": K,
            "warmup": args.warmup,
            "prime_steps": prime,  # untimed steps BEFORE the contracted warmup (XG_BENCH_PRIME; allocator, code objects, clocks)
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("NOT BASELINE's shape (plumbing check): " if toy else "") + f"configs[1]: Grid.interp+Grid.diff along X (periodic) and Y (extend) on one "
                                   f"{NX}x{NY}x{nz} f64 C-grid record per GPU, HBM-resident",
                       "cells_per_step_per_gpu": len(OPS) * cells_per_op, "records_per_gpu": 1,
                       "sharding": "record axis, no data-path collective"},
            "achieved_GBps_whole_step": round(total_cells * BYTES_PER_CELL / elapsed / 1e9 / world, 1),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         # True: the two PMC child passes above ran now; False: replayed from the committed passes named in
                         # traffic_source (N > 1, --no-pmc, no rocprofv3, or a pass that failed)
                         "traffic_measured_in_run": measured_now, "traffic_over_algorithmic": round(traffic / alg_bytes, 4) if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                         "per_op_ms": {f"{fn}_{ax}": round(ms, 4) for (fn, ax), ms in zip(OPS, per_op_ms)}},
            "ranks": {"world_size": n_ranks, "backend": ("nccl (RCCL)" if ranks.backend == "nccl" else str(ranks.backend)) if dist is not None else "single process",
                      "per_rank_ms_per_step": per_rank_ms_per_step,
                      # load balance INSIDE this run (min / max over the ranks' times) -- not scaling efficiency: scaling
                      # against N = 1 is the driver's to compute from the per-N values
                      "rank_balance_min_over_max": round(min(per_rank_ms_per_step) / max(per_rank_ms_per_step), 4) if max(per_rank_ms_per_step) > 0 else None,
                      "placement": placement, "numa_bind": os.environ.get("XG_NUMA_BIND", "1") != "0"},
            "device_ms_per_step": {"min": round(step_ms[0], 4), "median": round(step_ms[len(step_ms) // 2], 4),
                                   "max": round(step_ms[-1], 4), "first": round(step_ms_in_order[0], 4),
                                   "slowest_step": int(np.argmax(step_ms_in_order))},
            # where the results live (DESIGN section 8 "Placement"): scattered = separately created 64 MiB physical
            # allocations behind one virtual range (xg_pool_alloc); pool_fallbacks > 0 = some result fell back to hipMalloc
            "result_buffers": dict(_result_buffer_stats(), scattered=os.environ.get("XG_SCATTER_OUT", "1") != "0"),
        }
        if world == 1 and not args.no_cpu_baseline:
            before = ranks.placement.get("affinity_before")
            if ranks.placement.get("bound") and before:  # the CPU baseline is timed on ALL the box's host cores again
                S.set_affinity_all_threads(S.parse_cpulist(before))
            line["cpu_baseline"] = cpu_baseline()
            line["parity_spot_check"] = spot_check(spot)
        if cfg_block is not None:
            if not args.no_cpu_baseline:  # the oracle is the checker of the slabs kept by the timed spans, nothing else
                check_configs(cfg_block, cfg_slabs)
            line["configs"] = cfg_block
        print(json.dumps(line), flush=True)
    ranks.close()


def _result_buffer_stats():
    try:
        from xgcm_amd import _hip

        return _hip.scatter_stats()
    except Exception as exc:  # noqa: BLE001
        return {"error": str(exc)[:100]}


if __name__ == "__main__":
    main()
