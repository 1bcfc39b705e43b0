#!/usr/bin/env python3
"""Kernel-level timing on one GPU (tuning aid, not the judged bench).

    python tools/microbench.py [--shape 75,2400,3600] [--reps 10] [--cases stencil,cumsum,...]

Prints one JSON line per case: median ms, Gcell/s, algorithmic GB/s and fraction of 8 TB/s.
Tunables are read by the library from the environment once per process (XG_SEG, XG_NT_STORE,
XG_NT_LOAD), so sweep them by running this script several times.
"""

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from xgcm_amd import device as D  # noqa: E402


def timeit(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="75,2400,3600")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cases", default="copy,stencil,metric,cumsum,reduce,vort")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--const", action="store_true", help="constant-valued field instead of random (DVFS / data-toggle check)")
    args = ap.parse_args()
    shape = tuple(int(s) for s in args.shape.split(","))
    cases = set(args.cases.split(","))
    nz, ny, nx = shape
    cells = nz * ny * nx
    tag = {k: v for k, v in sorted(os.environ.items()) if k.startswith("XG_") and k != "XG_HIP_LIB"}  # tunables of this run

    dt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    if args.dtype == "f32":
        tag["dtype"] = "f32"
        _syn = D.synthetic
        D.synthetic = lambda *a, **k: _syn(*a, **{**k, "dtype": dt})  # every synthetic array of this run in f32
    T = D.synthetic(shape, 2)
    if args.const:
        T = D.synthetic(shape, 2, 0, 0.0, 1.0)
        tag["data"] = "const"
    out = []

    def rec(name, ms, best, bytes_per_cell, ncell=cells):
        bytes_per_cell = bytes_per_cell * esz / 8
        gbs = ncell * bytes_per_cell / (ms * 1e-3) / 1e9
        row = {"case": name, "ms": round(ms, 4), "best_ms": round(best, 4), "gcell_s": round(ncell / ms / 1e6, 3),
               "GBps": round(gbs, 1), "frac_8TBps": round(gbs / 8000, 4), **tag}
        print(json.dumps(row), flush=True)
        out.append(row)

    if "copy" in cases:
        dst = torch.empty_like(T)
        ms, b = timeit(lambda: dst.copy_(T), args.reps)
        rec("torch_copy(ref ceiling)", ms, b, 16)
        del dst
    if "stencil" in cases:
        for op in ("diff", "interp"):
            ms, b = timeit(lambda: D.stencil1d(op, T, 2, 1, 0, "periodic"), args.reps)
            rec(f"{op}_X_periodic_c2l", ms, b, 16)
            ms, b = timeit(lambda: D.stencil1d(op, T, 1, 1, 0, "extend"), args.reps)
            rec(f"{op}_Y_extend_c2l", ms, b, 16)
        ms, b = timeit(lambda: D.stencil1d("diff", T, 0, 1, 0, "fill"), args.reps)
        rec("diff_Z_fill_c2l", ms, b, 16)
        ms, b = timeit(lambda: D.stencil1d("diff", T, 2, 1, 1, "extend"), args.reps)
        rec("diff_X_extend_c2outer(scalar path)", ms, b, 16)
    if "metric" in cases:
        dx = D.synthetic((1, ny, nx), 31, 0, 1000.0, 1000.0)
        dz = D.synthetic((nz, 1, 1), 33, 0, 1000.0, 1000.0)
        ms, b = timeit(lambda: D.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx), args.reps)
        rec("derivative_X(dx 2D)", ms, b, 16 + 8 / nz)
        ms, b = timeit(lambda: D.stencil1d("diff", T, 1, 1, 0, "extend", m_out=dx), args.reps)
        rec("derivative_Y(dy 2D)", ms, b, 16 + 8 / nz)
        ms, b = timeit(lambda: D.stencil1d("diff", T, 0, 1, 0, "fill", m_out=dz), args.reps)
        rec("derivative_Z(dz 1D)", ms, b, 16)
        ms, b = timeit(lambda: D.stencil1d("interp", T, 2, 1, 0, "periodic", m_in=dx, m_out=dx), args.reps)
        rec("interp_X_metric_weighted", ms, b, 16 + 16 / nz)
        ms, b = timeit(lambda: D.stencil1d("interp", T, 1, 1, 0, "extend", m_in=dx, m_out=dx), args.reps)
        rec("interp_Y_metric_weighted", ms, b, 16 + 16 / nz)
    if "scanY" in cases:  # the long march with few columns, alone (tunable sweeps)
        ms, b = timeit(lambda: D.cumsum1d(T, 1, 0, 1, 1, 0, "fill"), args.reps)
        rec("cumsum_Y_c2l_fill", ms, b, 16)
        ms, b = timeit(lambda: D.reduce1d(T, 1, None), args.reps)
        rec("sum_Y", ms, b, 8)
    if "scanZ" in cases:
        ms, b = timeit(lambda: D.cumsum1d(T, 0, 0, 1, 1, 0, "fill"), args.reps)
        rec("cumsum_Z_c2l_fill", ms, b, 16)
        dz = D.synthetic((nz, 1, 1), 33, 0, 1000.0, 1000.0)
        ms, b = timeit(lambda: D.reduce1d(T, 0, dz), args.reps)
        rec("integrate_Z(drF 1D)", ms, b, 8 + 8 / nz)
    if "cumsum" in cases:
        ms, b = timeit(lambda: D.cumsum1d(T, 0, 0, 1, 1, 0, "fill"), args.reps)
        rec("cumsum_Z_c2l_fill", ms, b, 16)
        ms, b = timeit(lambda: D.cumsum1d(T, 1, 0, 1, 1, 0, "fill"), args.reps)
        rec("cumsum_Y_c2l_fill", ms, b, 16)
        ms, b = timeit(lambda: D.cumsum1d(T, 2, 0, 1, 1, 0, "fill"), args.reps)
        rec("cumsum_X_c2l_fill(block scan)", ms, b, 16)
        ms, b = timeit(lambda: D.cumsum1d(T, 2, 0, 0, 1, 0, "fill"), args.reps)
        rec("cumsum_X_c2outer_fill(odd rows)", ms, b, 16)
    if "reduce" in cases:
        dz = D.synthetic((nz, 1, 1), 33, 0, 1000.0, 1000.0)
        ms, b = timeit(lambda: D.reduce1d(T, 0, dz), args.reps)
        rec("integrate_Z(drF 1D)", ms, b, 8 + 8 / nz)
        ms, b = timeit(lambda: D.reduce1d(T, 1, None), args.reps)
        rec("sum_Y", ms, b, 8)
        ms, b = timeit(lambda: D.reduce1d(T, 2, None), args.reps)
        rec("sum_X(wave per row)", ms, b, 8)
    if "generic" in cases:
        ms, b = timeit(lambda: D.pad_nd(T, {2: (1, 1)}, {2: "periodic"}, {}), args.reps)
        rec("pad_X(1,1)_periodic(generic k_pad)", ms, b, 16)
        ms, b = timeit(lambda: D.pad_nd(T, {1: (0, 1), 2: (2, 0)}, {1: "extend", 2: "fill"}, {2: 1.5}), args.reps)
        rec("pad_Y(0,1)+X(2,0)(generic k_pad)", ms, b, 16)
        ms, b = timeit(lambda: D.pad_nd(T, {2: (1, 0)}, {2: "periodic"}, {}), args.reps)
        rec("pad_X(1,0)_periodic: odd row length", ms, b, 16)
        ms, b = timeit(lambda: D.pad_nd(T, {1: (1, 1), 0: (1, 0)}, {1: "extend", 0: "fill"}, {0: 0.0}), args.reps)
        rec("pad_Y(1,1)+Z(1,0) rows untouched along X (k_pad_rows)", ms, b, 16)
        B = D.synthetic(shape, 9)
        ms, b = timeit(lambda: D.binary("mul", T, B), args.reps)
        rec("binary_mul_full(3 streams)", ms, b, 24)
        dx = D.synthetic((1, ny, nx), 31, 0, 1000.0, 1000.0)
        ms, b = timeit(lambda: D.binary("div", T, dx), args.reps)
        rec("binary_div_bcast2D", ms, b, 16 + 8 / nz)
        dz = D.synthetic((nz, 1, 1), 33, 0, 1000.0, 1000.0)
        ms, b = timeit(lambda: D.binary("mul", T, dz), args.reps)
        rec("binary_mul_bcast1D", ms, b, 16)
        del B
    if "layout" in cases:  # xg_copy_nd against torch's own copy kernels on the same views
        views = [("copy (rows, 16-B lanes)", lambda: T[:, 1:, :]), ("transpose (Z,Y,X)->(Z,X,Y) (LDS tiles)", lambda: T.permute(0, 2, 1)),
                 ("transpose (Z,Y,X)->(X,Y,Z)", lambda: T.permute(2, 1, 0)), ("transpose (Z,Y,X)->(Y,X,Z)", lambda: T.permute(1, 2, 0)),
                 ("every second column (gather)", lambda: T[:, :, ::2])]
        for name, mk in views:
            v = mk()
            nc = v.numel()
            ms, b = timeit(lambda: D.materialize(v), args.reps)
            rec("layout: " + name, ms, b, 16, ncell=nc)
            ms, b = timeit(lambda: v.contiguous(), args.reps)
            rec("  torch .contiguous() of the same view", ms, b, 16, ncell=nc)
        ms, b = timeit(lambda: D.flip(T, [2]), args.reps)
        rec("layout: flip X (reversed rows)", ms, b, 16)
        ms, b = timeit(lambda: T.flip(2), args.reps)
        rec("  torch .flip(2)", ms, b, 16)
        ms, b = timeit(lambda: D.flip(T, [0]), args.reps)
        rec("layout: flip Z", ms, b, 16)
        ms, b = timeit(lambda: T.flip(0), args.reps)
        rec("  torch .flip(0)", ms, b, 16)
    if "vort" in cases:
        U = D.synthetic(shape, 51)
        V = D.synthetic(shape, 52)
        A = D.synthetic((1, ny, nx), 53, 0, 1000.0, 1000.0)
        ms, b = timeit(lambda: D.vorticity(U, V, A, "fill", "fill"), args.reps)
        rec("vorticity_fused", ms, b, 24 + 8 / nz)

        def unfused():
            dv = D.stencil1d("diff", V, 2, 1, 0, "fill")
            du = D.stencil1d("diff", U, 1, 1, 0, "fill")
            return D.binary("div", D.binary("sub", dv, du), A)

        ms, b = timeit(unfused, max(3, args.reps // 2))
        rec("vorticity_unfused(4 kernels; fused-equivalent bytes)", ms, b, 24 + 8 / nz)


if __name__ == "__main__":
    main()
