#!/usr/bin/env python3
"""WHY does the same kernel on the same data run 1.71 ms on one pair of buffers and 2.0 ms on another?  (VERDICT r04 next #3:
tools/survey.py and tools/roofline_table.py disagree by 6-15 % on cumsum Z, derivative Y, the two-axis metric interp, cumint
X/Y -- never on diff X / Y -- both as medians over fresh placements.)

One process, P pairs of (input, output) buffers alive AT THE SAME TIME, so that time, clocks and process state are common
to all of them and only the placement differs:

  1. every pair is timed through the raw C ABI, round-robin, twice -- a pair's rate is a property of the pair (stable
     across rounds), not of when it ran;
  2. the SAME input buffers through `Grid` (the operator path: output from torch's allocator) -- the two tools' difference
     is reproduced or not on identical inputs;
  3. the pairs are re-timed after the arena experiment: ONE allocation made first, carved into aligned slices
     (`--arena`), which is how the library can make placement its own property;
  4. with `--pmc-marks` the launches are issued in a fixed order with marker dispatches in between, so that a
     `rocprofv3 --pmc ...` run of this script attributes counters (TLB misses, TCC stalls, per-channel requests) to a
     FAST and a SLOW pair: `python tools/placement_probe.py --pmc GROUP` re-executes itself under rocprofv3 and joins.

    python tools/placement_probe.py --op cumZ --placements 8
    python tools/placement_probe.py --op cumZ --placements 6 --pmc "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from xgcm_amd import DataArray, Dataset, Grid, _hip  # noqa: E402
from xgcm_amd import device as D  # noqa: E402

SHAPE = (75, 2400, 3600)


def abi_launch(lib, op, src, dst, shape, metric, stream):
    sh = _hip.i64(shape)
    if op == "cumZ":
        return lib.xg_cumsum1d_f64(src, dst, sh, 3, 0, 0, 1, 0, 1, 1, 0, _hip.BC["fill"], 0.0, None, None, None, None, stream)
    if op == "cumY":
        return lib.xg_cumsum1d_f64(src, dst, sh, 3, 1, 0, 1, 0, 1, 1, 0, _hip.BC["extend"], 0.0, None, None, None, None, stream)
    if op == "diffX":
        return lib.xg_stencil1d_f64(0, src, dst, sh, 3, 2, shape[2], 1, 0, _hip.BC["periodic"], 0.0, None, None, None, None, stream)
    if op == "diffY":
        return lib.xg_stencil1d_f64(0, src, dst, sh, 3, 1, shape[1], 1, 0, _hip.BC["extend"], 0.0, None, None, None, None, stream)
    if op == "sumY":  # read-only march along Y (dst holds the small result)
        return lib.xg_reduce1d_f64(src, dst, sh, 3, 1, 1, None, None, stream)
    if op == "sumZ":
        return lib.xg_reduce1d_f64(src, dst, sh, 3, 0, 1, None, None, stream)
    if op == "fill":  # write-only, one stream
        return lib.xg_fill_synthetic_f64(dst, shape[0] * shape[1] * shape[2], 7, 0, 1.0, -0.5, stream)
    if op == "dY":  # derivative Y: output metric (1, Y, X)
        st = _hip.i64([0, shape[2], 1])
        return lib.xg_stencil1d_f64(0, src, dst, sh, 3, 1, shape[1], 1, 0, _hip.BC["extend"], 0.0, None, None, metric, st, stream)
    raise SystemExit(f"unknown op {op}")


def time_abi(lib, op, x, y, metric, reps):
    st = torch.cuda.current_stream().cuda_stream
    mp = metric.data_ptr() if metric is not None else None
    for _ in range(2):
        _hip.check(abi_launch(lib, op, x.data_ptr(), y.data_ptr(), list(SHAPE), mp, st))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        _hip.check(abi_launch(lib, op, x.data_ptr(), y.data_ptr(), list(SHAPE), mp, st))
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def time_fn(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


class Raw:
    """a device buffer obtained straight from the HIP runtime (hipExtMallocWithFlags): only an address"""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def data_ptr(self):
        return self.ptr


def hip_runtime():
    import ctypes

    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    rt.hipExtMallocWithFlags.restype = ctypes.c_int
    return rt


def contiguous_alloc(rt, nbytes):
    """physically contiguous VRAM (hipDeviceMallocContiguous = 0x4): ONE PTE fragment run, the best the TLB can get"""
    import ctypes

    p = ctypes.c_void_p()
    rc = rt.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 0x4)
    if rc != 0 or not p.value:
        raise RuntimeError(f"hipExtMallocWithFlags(contiguous, {nbytes}) -> {rc}")
    return Raw(p.value, nbytes)


def make_grid():
    nz, ny, nx = SHAPE
    coords = {"XC": np.arange(nx) + 0.5, "XG": np.arange(nx) * 1.0, "YC": np.arange(ny) + 0.5, "YG": np.arange(ny) * 1.0,
              "Z": np.arange(nz) + 0.5, "Zl": np.arange(nz) * 1.0}
    dyC = DataArray(D.synthetic((ny, nx), 32, 0, 1000.0, 1000.0), ("YG", "XC"))
    grid = Grid(Dataset({"dyC": dyC}, coords),
                coords={"X": {"center": "XC", "left": "XG"}, "Y": {"center": "YC", "left": "YG"}, "Z": {"center": "Z", "left": "Zl"}},
                padding={"X": "periodic", "Y": "extend", "Z": "fill"}, metrics={("Y",): ["dyC"]}, autoparse_metadata=False)
    return grid, dyC


def grid_call(grid, op, T):
    if op == "cumZ":
        return grid.cumsum(T, "Z")
    if op == "cumY":
        return grid.cumsum(T, "Y")
    if op == "diffX":
        return grid.diff(T, "X")
    if op == "diffY":
        return grid.diff(T, "Y")
    if op == "dY":
        return grid.derivative(T, "Y")
    raise SystemExit(op)


def run(a):
    lib = _hip.load()
    n = SHAPE[0] * SHAPE[1] * SHAPE[2]
    nbytes = n * 8
    alg = 2.0 * nbytes
    frac = lambda ms: round(alg / (ms * 1e-3) / 8e12, 4)  # noqa: E731
    arena = None
    if a.arena:  # FIRST allocation of the process: one block, sliced at 2 MiB multiples
        arena = torch.empty(a.arena * 2 * (nbytes + (2 << 20)) + (4 << 20), dtype=torch.uint8, device="cuda")
    grid, dyC = make_grid()
    metric = dyC.data.reshape(1, SHAPE[1], SHAPE[2]) if "dY" in a.op.split(",") else None
    pairs, pads = [], []
    for r in range(a.placements):
        pads.append(torch.empty((r * 37 + 1) << 20, dtype=torch.uint8, device="cuda"))  # shifts what the allocator hands out next
        x = D.synthetic(SHAPE, 2)
        y = torch.empty(SHAPE, dtype=torch.float64, device="cuda")
        pairs.append((x, y))
    if arena is not None:
        base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()
        step = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        for r in range(a.arena):
            x = arena[base + (2 * r) * step: base + (2 * r) * step + nbytes].view(torch.float64).view(SHAPE)
            y = arena[base + (2 * r + 1) * step: base + (2 * r + 1) * step + nbytes].view(torch.float64).view(SHAPE)
            x.copy_(pairs[0][0])
            pairs.append((x, y))
    names = [f"fresh{r}" for r in range(a.placements)] + [f"arena{r}" for r in range(a.arena)]
    if a.contig:
        rt = hip_runtime()
        st = torch.cuda.current_stream().cuda_stream
        for r in range(a.contig):
            try:
                x, y = contiguous_alloc(rt, nbytes), contiguous_alloc(rt, nbytes)
            except RuntimeError as exc:
                print(json.dumps({"contiguous_alloc_failed": str(exc)}), flush=True)
                break
            _hip.check(lib.xg_fill_synthetic_f64(x.data_ptr(), n, 2, 0, 1.0, -0.5, st))
            pairs.append((x, y))
            names.append(f"contig{r}")
    torch.cuda.synchronize()
    if a.pmc_marks:
        # fixed order under the profiler: marker (k_fill_synthetic of 4096 * (1 + index) cells), then 3 launches of that pair
        st = torch.cuda.current_stream().cuda_stream
        mp = metric.data_ptr() if metric is not None else None
        for idx, (x, y) in enumerate(pairs):
            D.synthetic((4096 * (1 + idx),), 1)   # the marker comes FIRST: every launch up to the next marker is this pair's
            for _ in range(4):                     # (the first of the four warms the caches and is dropped by the join)
                _hip.check(abi_launch(lib, a.op, x.data_ptr(), y.data_ptr(), list(SHAPE), mp, st))
            torch.cuda.synchronize()
            print(json.dumps({"pair": names[idx], "index": idx}), flush=True)
        return
    if a.scatter:
        # outputs BUILT by xg_scatter_alloc (chunk MiB, groups, spacer GiB) against ordinary allocations, several operators
        import ctypes

        x = pairs[0][0]
        ops = a.op.split(",")
        plain = [y for _, y in pairs]
        line = {"scatter_vs_plain": ops, "plain_ms": {op: [round(time_abi(lib, op, x, y, metric, a.reps), 3) for y in plain] for op in ops}}
        print(json.dumps(line), flush=True)
        for spec in a.scatter.split(";"):
            chunk_mb, groups, spacer_gb = (float(v) for v in spec.split(","))
            p = ctypes.c_void_p()
            rc = lib.xg_scatter_alloc(ctypes.byref(p), nbytes, int(chunk_mb * (1 << 20)), int(groups), int(spacer_gb * (1 << 30)))
            if rc != 0:
                print(json.dumps({"scatter": spec, "error": _hip.last_error()}), flush=True)
                continue
            y = Raw(p.value, nbytes)
            res = {op: round(time_abi(lib, op, x, y, metric, a.reps), 3) for op in ops}
            # ... and as the INPUT (is reading from a scattered buffer any different?)
            _hip.check(lib.xg_fill_synthetic_f64(y.data_ptr(), n, 2, 0, 1.0, -0.5, torch.cuda.current_stream().cuda_stream))
            rin = {op: round(time_abi(lib, op, y, plain[0], metric, a.reps), 3) for op in ops if op != "fill"}
            print(json.dumps({"scatter": {"chunk_MiB": chunk_mb, "groups": int(groups), "spacer_GiB": spacer_gb}, "as_output_ms": res,
                              "as_input_ms": rin}), flush=True)
            torch.cuda.synchronize()
            _hip.check(lib.xg_scatter_free(p))
        return
    if a.alloc_cost:
        # what does a scattered buffer cost to MAKE (the pool pays it once per cached block)?
        import ctypes
        import time

        for gb in (1, 5, 13):
            size = gb << 30
            for label, chunk in (("hipMalloc (torch.empty, allocator cache emptied)", 0), ("scatter 64 MiB", 64), ("scatter 256 MiB", 256)):
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                t0 = time.perf_counter()
                if chunk == 0:
                    buf = torch.empty(size, dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    del buf
                    torch.cuda.empty_cache()
                    t2 = time.perf_counter()
                else:
                    p = ctypes.c_void_p()
                    _hip.check(lib.xg_scatter_alloc(ctypes.byref(p), size, chunk << 20, 1, 0))
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    _hip.check(lib.xg_scatter_free(p))
                    t2 = time.perf_counter()
                print(json.dumps({"alloc_cost": label, "GiB": gb, "alloc_ms": round((t1 - t0) * 1e3, 2), "free_ms": round((t2 - t1) * 1e3, 2)}), flush=True)
        return
    if a.matrix:
        # 2 x 2: input and output each from an ordinary allocation or from xg_scatter_alloc (chunk MiB), several operators
        import ctypes

        ops = a.op.split(",")
        st = torch.cuda.current_stream().cuda_stream

        def scat():
            p = ctypes.c_void_p()
            _hip.check(lib.xg_scatter_alloc(ctypes.byref(p), nbytes, int(a.matrix) << 20, 1, 0))
            return Raw(p.value, nbytes)

        plain_in, plain_out = pairs[0]
        sc_in, sc_out = scat(), scat()
        _hip.check(lib.xg_fill_synthetic_f64(sc_in.data_ptr(), n, 2, 0, 1.0, -0.5, st))
        for rnd in range(2):
            res = {}
            for op in ops:
                res[op] = {"plain->plain": round(time_abi(lib, op, plain_in, plain_out, metric, a.reps), 3),
                           "plain->scatter": round(time_abi(lib, op, plain_in, sc_out, metric, a.reps), 3),
                           "scatter->plain": round(time_abi(lib, op, sc_in, plain_out, metric, a.reps), 3),
                           "scatter->scatter": round(time_abi(lib, op, sc_in, sc_out, metric, a.reps), 3)}
            print(json.dumps({"matrix_chunk_MiB": a.matrix, "round": rnd, "ms": res}), flush=True)
        return
    if a.sweep:
        # the output base slid through ONE allocation in fixed steps, the input fixed: is the rate a function of the address?
        gb, step_mb = a.sweep
        span = int(gb) << 30
        raw = None
        if a.sweep_contig:
            raw = contiguous_alloc(hip_runtime(), span + nbytes + (4 << 20))
            base = (raw.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        else:
            big = torch.empty(span + nbytes + (4 << 20), dtype=torch.uint8, device="cuda")
            base = (big.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        x = pairs[0][0]
        out = []
        off = 0
        while off <= span:
            y = Raw(base + off, nbytes)
            ms = time_abi(lib, a.op, x, y, metric, 3)
            out.append((off >> 20, round(ms, 3)))
            off += int(step_mb) << 20
        print(json.dumps({"sweep": a.op, "contiguous": bool(a.sweep_contig), "base": hex(base), "step_MiB": int(step_mb),
                          "offset_MiB__ms": out}), flush=True)
        return
    if a.cross:
        # which buffer's placement matters?  time every input against every output
        k = min(a.cross, len(pairs))
        mat = [[round(time_abi(lib, a.op, pairs[i][0], pairs[j][1], metric, a.reps), 3) for j in range(k)] for i in range(k)]
        print(json.dumps({"cross": a.op, "rows_are_inputs_cols_are_outputs": names[:k], "ms": mat}), flush=True)
        return
    rounds = []
    for rnd in range(2):
        rounds.append([time_abi(lib, a.op, x, y, metric, a.reps) for x, y in pairs])
    for idx, (x, y) in enumerate(pairs):
        line = {"op": a.op, "pair": names[idx], "in_ptr": hex(x.data_ptr()), "out_ptr": hex(y.data_ptr()),
                "abi_ms_round1": round(rounds[0][idx], 4), "abi_ms_round2": round(rounds[1][idx], 4),
                "abi_frac": frac(min(rounds[0][idx], rounds[1][idx]))}
        if isinstance(x, torch.Tensor):
            T = DataArray(x, ("Z", "YC", "XC"), name="T")
            g = time_fn(lambda: grid_call(grid, a.op, T), a.reps)
            out = grid_call(grid, a.op, T).data   # (the operator path allocates its own output)
            line.update({"grid_ms_own_output": round(g, 4), "grid_frac": frac(g), "grid_out_ptr": hex(out.data_ptr())})
            del out
        print(json.dumps(line), flush=True)
    ab = [min(r0, r1) for r0, r1 in zip(*rounds)]
    print(json.dumps({"summary": a.op, "abi_ms_min": round(min(ab), 4), "abi_ms_max": round(max(ab), 4),
                      "spread_pct": round(100 * (max(ab) / min(ab) - 1), 2),
                      "round_to_round_max_pct": round(100 * max(abs(r0 / r1 - 1) for r0, r1 in zip(*rounds)), 2)}), flush=True)


def with_pmc(a, argv):
    """this script (--pmc-marks) under `rocprofv3 --pmc <counters> --kernel-trace`; counters per pair, instances summed and,
    where the tool recorded dimensions, listed per instance (per-channel skew)"""
    tmp = tempfile.mkdtemp(prefix="placement_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + a.pmc.replace(",", " ").split() + ["--kernel-trace", "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                                                   "--pmc-marks"] + argv
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        print(json.dumps({"error": r.stdout[-1500:]}), flush=True)
        return
    con = sqlite3.connect(dbs[0])
    tables = [t[0] for t in con.execute("select name from sqlite_master where type in ('table','view')")]
    def tab(prefix):
        c = [t for t in tables if t.startswith(prefix)]
        return c[0] if c else None
    out = {"tables": [t for t in tables if "pmc" in t or "counter" in t][:12]}
    # kernels in launch order with their dispatch ids; counters joined through the views rocprofv3 ships
    try:
        rows = list(con.execute("select dispatch_id, name, grid_x, duration from kernels order by start"))
    except sqlite3.OperationalError:
        rows = []
    cur, groups = None, {}
    for did, name, gx, dur in rows:
        if "k_fill_synthetic" in name and gx % 4096 == 0 and gx < 4096 * 512 and gx > 0:
            cur = gx // 4096 - 1
            continue
        if cur is not None and "k_fill_synthetic" not in name:
            groups.setdefault(cur, []).append((did, dur / 1e3))
    groups = {k: v[1:] for k, v in groups.items()}  # (the first launch after a marker is the warm-up)
    # the raw event tables keep one row per counter INSTANCE (channel, shader engine ...) where the view above sums them
    try:
        cols = {t: [c[1] for c in con.execute(f"pragma table_info({t})")] for t in ("pmc_events", "pmc_info") if t in tables}
        out_schema = {t: c for t, c in cols.items()}
    except sqlite3.OperationalError:
        out_schema = {}
    counters = {}
    try:
        q = ("select dispatch_id, counter_name, sum(value), max(value), count(*) from counters_collection "
             "group by dispatch_id, counter_name")
        for did, cname, val, vmax, nrow in con.execute(q):
            counters.setdefault(did, {})[cname] = val
            if nrow > 1 and val:  # one row per instance (channel / SE ...): how far the busiest one is above the mean
                counters[did][cname + ":max_over_mean"] = vmax / (val / nrow)
                counters[did][cname + ":instances"] = nrow
    except sqlite3.OperationalError as exc:
        out["counter_query_error"] = str(exc)
    out["schema"] = out_schema
    print(json.dumps(out), flush=True)
    if a.dump_db:
        shutil.copy(dbs[0], a.dump_db)
    for ln in lines:
        g = groups.get(ln["index"], [])
        ln["trace_us"] = [round(d, 1) for _, d in g]
        agg = {}
        for did, _ in g:
            for k, v in counters.get(did, {}).items():
                agg.setdefault(k, []).append(v)
        ln["counters_per_launch"] = {k: round(sum(v) / len(v), 3) for k, v in agg.items()}
        print(json.dumps(ln), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="cumZ")
    ap.add_argument("--placements", type=int, default=6)
    ap.add_argument("--arena", type=int, default=0, help="also time this many pairs carved out of ONE allocation made first")
    ap.add_argument("--contig", type=int, default=0, help="also time this many pairs of physically CONTIGUOUS buffers (hipExtMallocWithFlags)")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--pmc", default="", help="counter names (comma- or space-separated): run under rocprofv3 and join per pair")
    ap.add_argument("--pmc-marks", action="store_true")
    ap.add_argument("--cross", type=int, default=0, help="time input i against output j for the first N pairs (which buffer's placement matters?)")
    ap.add_argument("--sweep", type=float, nargs=2, default=None, metavar=("GiB", "STEP_MiB"),
                    help="slide the OUTPUT base through one allocation of GiB in steps of STEP_MiB (input fixed)")
    ap.add_argument("--sweep-contig", action="store_true", help="the swept allocation is physically contiguous")
    ap.add_argument("--scatter", default="", help="';'-separated 'chunk_MiB,groups,spacer_GiB': time --op (comma list) into outputs built by xg_scatter_alloc")
    ap.add_argument("--matrix", type=int, default=0, metavar="CHUNK_MiB", help="input x output, each plain or scattered (chunk size), for the --op list")
    ap.add_argument("--alloc-cost", action="store_true", help="time making / releasing plain and scattered buffers of 1, 5, 13 GiB")
    ap.add_argument("--dump-db", default="", help="with --pmc: keep rocprofv3's database at this path")
    a = ap.parse_args()
    if a.pmc:
        argv = []
        skip = False
        for x in sys.argv[1:]:
            if skip:
                skip = False
                continue
            if x in ("--pmc", "--dump-db"):
                skip = True
                continue
            argv.append(x)
        with_pmc(a, argv)
        return
    run(a)


if __name__ == "__main__":
    main()
