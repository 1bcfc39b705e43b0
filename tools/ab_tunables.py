#!/usr/bin/env python3
"""Interleaved A/B of launch-shape tunables on one GPU, inside ONE process (xg_set_tunable).

    python tools/ab_tunables.py --cases dX,dY,iXmw --variants "contig_rw=0,met_seg=1;contig_rw=2,met_seg=2" [--rounds 6]

A device drifts by several percent while it warms up, so variants timed one process after the other cannot be
compared; here every round times every variant once (order rotated per round) and the table reports the median
over the rounds.  Cases: the full-size 75 x 2400 x 3600 f64 operators of tools/microbench.py.  Tuning aid only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from xgcm_amd import _hip  # noqa: E402
from xgcm_amd import device as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="dX,dY,dZ,iXmw,iYmw")
    ap.add_argument("--variants", required=True, help="';'-separated variants, each 'name=value,name=value'")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shape", default="75,2400,3600")
    ap.add_argument("--mark", action="store_true", help="a marker dispatch (k_fill_synthetic of 4096 * (1 + variant * ncases + case) "
                    "cells) before every (variant, case) block: tools/pmc_ab.py maps rocprofv3 dispatches to variants with it")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="f32: every synthetic operand in float32 (bytes per cell halve)")
    ap.add_argument("--realloc", action="store_true", help="every round on FRESH buffers (allocator cache emptied, the next allocations "
                    "shifted): a kernel's rate depends on where the driver placed its buffers -- cumsum Z 1.71 to 2.03 ms on one box, "
                    "profiles/history/r04a_addr_probe.jsonl -- so a table that is to agree with another process's must report the median over placements")
    a = ap.parse_args()
    bscale = 1.0
    if a.dtype == "f32":
        import functools
        D.synthetic = functools.partial(D.synthetic, dtype=torch.float32)
        bscale = 0.5
    nz, ny, nx = (int(v) for v in a.shape.split(","))
    cells = nz * ny * nx
    T = D.synthetic((nz, ny, nx), 2)
    dx = D.synthetic((1, ny, nx), 31, 0, 1000.0, 1000.0)
    dx2 = D.synthetic((1, ny, nx), 32, 0, 1000.0, 1000.0)
    dz = D.synthetic((nz, 1, 1), 33, 0, 1000.0, 1000.0)
    dy1 = D.synthetic((1, ny, 1), 34, 0, 1000.0, 1000.0)
    U = V = None
    To = D.synthetic((nz, ny, nx + 1), 3) if "diffYo" in a.cases.split(",") else None
    CASES = {
        "diffX": (lambda: D.stencil1d("diff", T, 2, 1, 0, "periodic"), 16),
        "diffY": (lambda: D.stencil1d("diff", T, 1, 1, 0, "extend"), 16),
        "dX": (lambda: D.stencil1d("diff", T, 2, 1, 0, "periodic", m_out=dx), 16 + 8 / nz),
        "dY": (lambda: D.stencil1d("diff", T, 1, 1, 0, "extend", m_out=dx), 16 + 8 / nz),
        "dZ": (lambda: D.stencil1d("diff", T, 0, 1, 0, "fill", m_out=dz), 16),
        "diffZ": (lambda: D.stencil1d("diff", T, 0, 1, 0, "extend"), 16),
        "diffXo": (lambda: D.stencil1d("diff", T, 2, 1, 1, "extend"), 16),   # center -> outer: rows of N + 1 cells (K1g)
        "diffYo": (lambda: D.stencil1d("diff", To, 1, 1, 0, "extend"), 16),  # rows of 3601 cells along Y (K2g)
        "dY3": (lambda: D.stencil1d("diff", T, 1, 1, 0, "extend", m_out=T2), 24),
        "padX": (lambda: D.pad_nd(T, {2: (1, 1)}, {2: "periodic"}, {}), 16),
        "padYX": (lambda: D.pad_nd(T, {1: (0, 1), 2: (2, 0)}, {1: "extend", 2: "fill"}, {2: 1.5}), 16),
        "padYZ": (lambda: D.pad_nd(T, {1: (1, 1), 0: (1, 0)}, {1: "extend", 0: "fill"}, {0: 0.0}), 16),
        "i2": (lambda: D.stencil2d("interp", T, 0, (1, 0), "periodic", 0.0, (1, 0), "extend", 0.0), 16),
        "i2mw": (lambda: D.stencil2d("interp", T, 0, (1, 0), "periodic", 0.0, (1, 0), "extend", 0.0, metrics=(dx[0], dx2[0], dx[0])), 16 + 24 / nz),
        "iZmw": (lambda: D.stencil1d("interp", T, 0, 1, 0, "fill", m_in=dz, m_out=dz), 16),
        "iYmw1": (lambda: D.stencil1d("interp", T, 1, 1, 0, "extend", m_in=dy1, m_out=dy1), 16),
        "iXmw": (lambda: D.stencil1d("interp", T, 2, 1, 0, "periodic", m_in=dx2, m_out=dx), 16 + 16 / nz),
        "iYmw": (lambda: D.stencil1d("interp", T, 1, 1, 0, "extend", m_in=dx2, m_out=dx), 16 + 16 / nz),
        "divT": (lambda: D.binary("div", T, dx), 16 + 8 / nz),
        "mulT": (lambda: D.binary("mul", T, dx), 16 + 8 / nz),   # the same streams with a product instead of a quotient: what the IEEE division costs
        "mulTT": (lambda: D.binary("mul", T, T2), 24),
        "cumY": (lambda: D.cumsum1d(T, 1, 0, 1, 1, 0, "fill"), 16),
        "cumZ": (lambda: D.cumsum1d(T, 0, 0, 1, 1, 0, "fill"), 16),
        "cumX": (lambda: D.cumsum1d(T, 2, 0, 1, 1, 0, "fill"), 16),
        "sumY": (lambda: D.reduce1d(T, 1, None), 8),
        "sumZ": (lambda: D.reduce1d(T, 0, dz), 8 + 8 / nz),
        "sumX": (lambda: D.reduce1d(T, 2, None), 8),
        "sumXw": (lambda: D.reduce1d(T, 2, dx), 8 + 8 / nz),
        "sumYw": (lambda: D.reduce1d(T, 1, dx), 8 + 8 / nz),
        "avgYw": (lambda: D.reduce1d(T, 1, dx, "mean_valid"), 8 + 8 / nz),  # Grid.average along Y with dy(Y, X)
        "sumYw1": (lambda: D.reduce1d(T, 1, dy1), 8),
        "sumYw3": (lambda: D.reduce1d(T, 1, T3), 16),
        "cumYw": (lambda: D.cumsum1d(T, 1, 0, 0, 0, 0, None, 0.0, False, True, dx, None), 16 + 8 / nz),
        "cumXp": (lambda: D.cumsum1d(T, 2, 0, 1, 1, 0, "periodic"), 16),
        "cumXe": (lambda: D.cumsum1d(T, 2, 0, 1, 1, 0, "extend"), 16),
        "cumX0": (lambda: D.cumsum1d(T, 2, 0, 0, 0, 0, None), 16),
        "cumXw": (lambda: D.cumsum1d(T, 2, 0, 0, 0, 0, None, 0.0, False, True, dx, None), 16 + 8 / nz),
        "cumXwl": (lambda: D.cumsum1d(T, 2, 0, 1, 1, 0, "fill", 0.0, False, True, dx, None), 16 + 8 / nz),  # Grid.cumint along X
        "cumZw": (lambda: D.cumsum1d(T, 0, 0, 0, 0, 0, None, 0.0, False, True, dx, None), 16 + 8 / nz),
        "sumZw2": (lambda: D.reduce1d(T, 0, dx), 8 + 8 / nz),
        "vort": (lambda: D.vorticity(U, V, dx, "fill", "fill"), 24 + 8 / nz),
        "divg": (lambda: D.divergence(U, V, dx, "periodic", "extend"), 24 + 8 / nz),
        "grad": (lambda: D.gradient(T, "periodic", "extend", 0.0, 0.0, dx, dx2), 24 + 16 / nz),
        "grad0": (lambda: D.gradient(T, "periodic", "extend", 0.0, 0.0, None, None), 24),  # one stream in, two out, no metric: the shape's ceiling
        "grad1": (lambda: D.gradient(T, "periodic", "extend", 0.0, 0.0, dx, None), 24 + 8 / nz),
        "grad_same": (lambda: D.gradient(T, "periodic", "extend", 0.0, 0.0, dx, dx), 24 + 8 / nz),      # two metric loads per row, ONE plane
        "grad_off": (lambda: D.gradient(T, "periodic", "extend", 0.0, 0.0, dx, dx2_off), 24 + 16 / nz),  # the second plane 2 KiB + 256 B further on
        "flux": (lambda: D.flux(U, V, T, "periodic", "extend"), 40),
    }
    cases = a.cases.split(",")
    dx2_off = None
    if "grad_off" in cases:
        pad = (2048 + 256) // dx2.element_size()
        big = torch.empty(ny * nx + pad, dtype=dx2.dtype, device="cuda")
        dx2_off = big[pad:].view(1, ny, nx)
        dx2_off.copy_(dx2)
    T2 = D.synthetic((nz, ny, nx), 9, 0, 1000.0, 1000.0) if ("mulTT" in cases or "dY3" in cases) else None
    dy1 = D.synthetic((1, ny, 1), 35, 0, 1000.0, 1000.0)
    T3 = D.synthetic((nz, ny, nx), 10, 0, 1000.0, 1000.0) if "sumYw3" in cases else None
    if any(c.startswith("t") and c[1:4] in ("lin", "con") for c in cases):
        # vertical transform of the field onto 50 levels / bins (tools/bench_configs.py --configs f4): theta =
        # running sum of positive random increments ("rw") or a smooth stratification ("sm")
        mt = 50
        inc = D.synthetic((nz, ny, nx), 72, 0, 1.0, 0.55)
        th_rw = D.cumsum1d(inc, 0, 0, 0, 0, 0, None, 0.0, False, False)
        inc_o = D.synthetic((nz + 1, ny, nx), 73, 0, 1.0, 0.55)
        tho_rw = D.cumsum1d(inc_o, 0, 0, 0, 0, 0, None, 0.0, False, False)
        del inc, inc_o
        import numpy as np
        zz = torch.arange(nz + 1, dtype=torch.float64, device="cuda")[:, None, None]
        yy = torch.arange(ny, dtype=torch.float64, device="cuda")[None, :, None]
        xx = torch.arange(nx, dtype=torch.float64, device="cuda")[None, None, :]
        wave2 = 2.0 * torch.sin(2 * np.pi * xx / nx) * torch.cos(2 * np.pi * yy / ny)
        th_sm = (1.05 * (zz[:nz] + 0.5) + wave2).contiguous()
        tho_sm = (1.05 * zz + wave2).contiguous()
        del zz, yy, xx, wave2
        levels = torch.linspace(1.0, 0.9 * nz, mt, dtype=torch.float64, device="cuda").reshape(mt, 1, 1)
        edges = torch.linspace(0.0, 1.6 * (nz + 1), mt + 1, dtype=torch.float64, device="cuda")
        if a.dtype == "f32":  # coordinates of a float32 field are float32 too (MITgcm / LLC4320 output)
            th_sm, tho_sm, levels, edges = th_sm.float(), tho_sm.float(), levels.float(), edges.float()
        CASES.update({
            "tlin_rw": (lambda: D.transform_linear(T, th_rw, levels, 0), (2 * nz + mt) * 8 / nz),
            "tlin_sm": (lambda: D.transform_linear(T, th_sm, levels, 0), (2 * nz + mt) * 8 / nz),
            "tcon_rw": (lambda: D.transform_conservative(T, tho_rw, edges, 0), (2 * nz + 1 + mt) * 8 / nz),
            "tcon_sm": (lambda: D.transform_conservative(T, tho_sm, edges, 0), (2 * nz + 1 + mt) * 8 / nz),
        })
    if "gatherYX" in cases:  # halo gather through a token map: periodic frame of one cell around (Y, X), every level
        import numpy as np
        idx = np.arange(ny * nx, dtype=np.int64).reshape(ny, nx)
        tok = (np.pad(idx, ((1, 1), (1, 1)), mode="wrap") + 1).reshape(-1)
        tokd = torch.from_numpy(tok).cuda()
        CASES["gatherYX"] = (lambda: D.gather(T, None, tokd, (False, True, True), (0, 1, 1), (nz, ny + 2, nx + 2), [0.0]), 16)
    for c in cases:  # cumsum along Z of R records in ONE launch ("cumZr4": R = 4): what does a launch cost beyond its bytes?
        if c.startswith("cumZr"):
            R = int(c[5:])
            TR = D.synthetic((R, nz, ny, nx), 4)
            CASES[c] = ((lambda TR=TR: D.cumsum1d(TR, 1, 0, 1, 1, 0, "fill")), 16 * R)
    if any(c in cases for c in ("vort", "divg", "flux")):
        U, V = D.synthetic((nz, ny, nx), 51), D.synthetic((nz, ny, nx), 52)
    variants = []
    for spec in a.variants.split(";"):
        kv = dict((k.strip(), int(v)) for k, v in (item.split("=") for item in spec.split(",") if item.strip()))
        variants.append(kv)
    defaults = {k: _hip.get_tunable(k) for kv in variants for k in kv}
    times = {(vi, c): [] for vi in range(len(variants)) for c in cases}
    for c in cases:  # warm-up: code objects, allocator, clocks
        for _ in range(3):
            CASES[c][0]()
    torch.cuda.synchronize()
    _shift = None
    for rnd in range(a.rounds):
        if a.realloc and rnd > 0:  # the big full-size operands on fresh allocations (small metrics and transform tables stay)
            _shift = None
            T = U = V = None
            T2k, T3k = T2 is not None, T3 is not None
            T2 = T3 = None
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            _shift = torch.empty((rnd * 37 + 1) << 20, dtype=torch.uint8, device="cuda")
            if a.mark:  # a marker that belongs to no (variant, case) block: tools/pmc_ab.py drops what follows it
                D.synthetic((4096 * (1 + len(variants) * len(cases)),), 1)
            T = D.synthetic((nz, ny, nx), 2)
            if T2k:
                T2 = D.synthetic((nz, ny, nx), 9, 0, 1000.0, 1000.0)
            if T3k:
                T3 = D.synthetic((nz, ny, nx), 10, 0, 1000.0, 1000.0)
            if any(c in cases for c in ("vort", "divg", "flux")):
                U, V = D.synthetic((nz, ny, nx), 51), D.synthetic((nz, ny, nx), 52)
            for c in cases:
                CASES[c][0]()
            torch.cuda.synchronize()
        order = list(range(len(variants)))
        order = order[rnd % len(order):] + order[:rnd % len(order)]
        for vi in order:
            for k, v in defaults.items():
                _hip.set_tunable(k, v)
            for k, v in variants[vi].items():
                _hip.set_tunable(k, v)
            for ci, c in enumerate(cases):
                fn = CASES[c][0]
                if a.mark:
                    D.synthetic((4096 * (1 + vi * len(cases) + ci),), 1)
                fn()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
                ev[0].record()
                for i in range(a.reps):
                    fn()
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps))
                times[(vi, c)].append(ts[len(ts) // 2])
    for vi, kv in enumerate(variants):
        for c in cases:
            ts = sorted(times[(vi, c)])
            ms = ts[len(ts) // 2]
            gbs = cells * CASES[c][1] * bscale / (ms * 1e-3) / 1e9
            # paired statistic: within a round every variant ran on the same buffers (the rate of a kernel depends on where
            # the driver placed them), so the ratio to the FIRST variant round by round is far tighter than the two medians
            ratios = sorted(a_ / b_ for a_, b_ in zip(times[(vi, c)], times[(0, c)]) if b_ > 0)
            print(json.dumps({"case": c, "variant": ",".join(f"{k}={v}" for k, v in kv.items()), "median_ms": round(ms, 4),
                              "time_vs_first_paired": round(ratios[len(ratios) // 2], 4) if ratios else None,
                              "min_ms": round(ts[0], 4), "max_ms": round(ts[-1], 4), "GBps": round(gbs, 1),
                              "frac_8TBps": round(gbs / 8000, 4), "rounds": a.rounds,
                              "alg_bytes": int(round(cells * CASES[c][1] * bscale))}), flush=True)


if __name__ == "__main__":
    main()
