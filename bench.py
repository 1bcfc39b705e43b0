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
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 16.0  # 1 f64 read + 1 f64 write per output cell
OPS = [("interp", "X"), ("diff", "X"), ("interp", "Y"), ("diff", "Y")]


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
    stand-in for the reference's dask threaded scheduler over a field chunked along Z."""
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
                       "host_cores": os.cpu_count(), "numpy": np.__version__,
                       "sample": f"{rounds} round(s), one 1x{NY}x{NX} level per task on {nthreads} threads "
                                 f"(all {os.cpu_count()} host cores unless host memory bounds it), {el:.1f} s"}
    return out


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
                   "--steps", "3", "--warmup", "1", "--levels", str(levels), "--no-cpu-baseline", "--no-pmc"]
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
            vals = [v for name, v in rows if dominant in name and v * 1024 > 0.2 * alg_bytes]
            if not vals:
                return None, f"no {dominant} dispatch in the {counter} pass", time.perf_counter() - t0
            kib[counter] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return kib["FETCH_SIZE"] * 1024 * 2 + kib["WRITE_SIZE"] * 1024, "measured", time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--levels", type=int, default=NZ, help="Z levels (default = the full 75; smaller only for debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child passes (N = 1 only; "
                    "also XG_BENCH_PMC=0): the committed figure of profiles/pmc_traffic.json is reported instead")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from xgcm_amd import sharding as S

    # `--gpus N` means N ranks whoever started us: under torchrun we ARE one of them (WORLD_SIZE must agree),
    # started by hand we re-execute under torch.distributed.run; fewer visible GPUs than ranks is an error
    S.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    ranks = S.init_ranks(args.gpus, backend="nccl")
    world, rank, local_rank, dist = ranks.world, ranks.rank, ranks.local_rank, ranks.dist

    import __graft_entry__ as entry

    if entry._stale():  # clean checkout: the .so is a git-ignored build artefact
        if rank == 0:
            entry.build()
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
    from xgcm_amd import device as D

    nz = args.levels
    cells_per_op = nz * NY * NX
    # each rank generates ITS record (seed 2, disjoint index range) directly in HBM
    field = D.synthetic((nz, NY, NX), 2, offset=rank * cells_per_op)
    grid, T = build_grid(nz, field)

    def step(events=None):
        for i, (fn, ax) in enumerate(OPS):
            getattr(grid, fn)(T, ax)
            if events is not None:
                events[i + 1].record()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

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
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    K = args.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(OPS) + 1)] for _ in range(K)]
    import gc

    gc.collect()
    gc.disable()  # an interpreter GC pause (~10 ms, seen once in r01d) is not part of the hot path
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record()
        step(ev[k])
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()

    local_cells = K * len(OPS) * cells_per_op
    per_rank_ms_per_step = [round(v / K * 1e3, 4) for v in ranks.gather_floats(elapsed)]
    # what a < 7x curve would have to be explained with: where each rank ran (GPU, NUMA node, CPUs it was bound to) and how
    # much of its wall time per step was NOT device time (Python dispatch of the 4 launches, barrier skew)
    local_dev_ms = sorted(ev[k][0].elapsed_time(ev[k][len(OPS)]) for k in range(K))
    placement = ranks.gather_objects(dict(ranks.placement, rank=rank,
                                          device_ms_per_step=round(local_dev_ms[len(local_dev_ms) // 2], 4),
                                          host_overhead_ms_per_step=round(elapsed / K * 1e3 - float(np.mean(local_dev_ms)), 4)))
    cells_per_s, elapsed = S.whole_job_throughput(local_cells, elapsed, dist, "cuda")
    n_ranks = dist.get_world_size() if dist is not None else 1  # the rank count RCCL itself reports

    # per-launch durations from the HIP events recorded on the launch stream inside the timed region
    per_op_ms = [float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(K)])) for i in range(len(OPS))]
    step_ms = sorted(ev[k][0].elapsed_time(ev[k][len(OPS)]) for k in range(K))  # device time per step
    by_kernel, of_axis = {}, kernel_of_axis()
    for (fn, ax), ms in zip(OPS, per_op_ms):
        by_kernel.setdefault(of_axis[ax], []).append(ms)
    dominant = max(by_kernel, key=lambda k: sum(by_kernel[k]))
    dom_ms = float(np.mean(by_kernel[dominant]))
    alg_bytes = cells_per_op * BYTES_PER_CELL
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9

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
        if traffic is None and os.path.exists(pmc):
            try:
                table = json.load(open(pmc))
                traffic = table.get(dominant)
                if traffic is not None:
                    traffic_source = (traffic_source or "") + "offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " + \
                        str(table.get("_source", "profiles/pmc_traffic.json"))
            except Exception:
                traffic = None
        line = {
            "metric": "stencil-cells/s, Grid.interp+Grid.diff (X periodic, Y extend) on 3600x2400x75 f64",
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
            "config": {"workload": f"configs[1]: Grid.interp+Grid.diff along X (periodic) and Y (extend) on one "
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
            "ranks": {"world_size": n_ranks, "backend": "nccl (RCCL)" if dist is not None else "single process",
                      "per_rank_ms_per_step": per_rank_ms_per_step,
                      # load balance INSIDE this run (min / max over the ranks' times) -- not scaling efficiency: scaling
                      # against N = 1 is the driver's to compute from the per-N values
                      "rank_balance_min_over_max": round(min(per_rank_ms_per_step) / max(per_rank_ms_per_step), 4) if max(per_rank_ms_per_step) > 0 else None,
                      "placement": placement, "numa_bind": os.environ.get("XG_NUMA_BIND", "1") != "0"},
            "device_ms_per_step": {"min": round(step_ms[0], 4), "median": round(step_ms[len(step_ms) // 2], 4),
                                   "max": round(step_ms[-1], 4)},
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
