#!/usr/bin/env python3
"""Hardware counters per (variant, case) of tools/ab_tunables.py, one rocprofv3 --pmc pass per counter group.

    python tools/pmc_ab.py --cases vort,mulTT --variants "vec_nt=0;vec_nt=3" --pmc "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum"

Every pass runs `ab_tunables.py --mark --rounds 1` under `rocprofv3 --pmc <group> --kernel-trace` (never with the
sys / hip / hsa trace domains: gpurun refuses that combination); the marker dispatches ab_tunables emits before every
(variant, case) block tell which dispatches belong to which variant.  Output, printed as soon as a pass ends: one JSON line
per (pass, variant, case, kernel) with the median duration under the profiler and the mean of every counter over the full-size dispatches of the block (the first
dispatch of a block is dropped: it runs right behind the marker, on cold caches).  FETCH_SIZE / WRITE_SIZE are printed
raw (KiB) and as bytes (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md "HBM")."""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", required=True)
    ap.add_argument("--variants", required=True)
    ap.add_argument("--pmc", required=True, help="'|'-separated counter groups, each a space-separated list")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shape", default="75,2400,3600")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--min-us", type=float, default=200.0, help="dispatches shorter than this are not reported")
    ap.add_argument("--pass-timeout", type=int, default=150, help="seconds one rocprofv3 pass may take")
    ap.add_argument("--placements", type=int, default=1, help="rounds on fresh buffers (ab_tunables --realloc): the reported duration is "
                    "the median over all of them")
    a = ap.parse_args()
    cases = a.cases.split(",")
    variants = a.variants.split(";")
    ap_timeout = a.pass_timeout
    for group in a.pmc.split("|"):
        group = group.strip()
        table = {}  # (vi, ci, kernel) -> {"dur": [...], counter: [...]}
        tmp = tempfile.mkdtemp(prefix="pmcab_", dir="/tmp")
        # the group "TRACE" is a plain kernel-trace pass: durations without any counter collected
        cmd = ["rocprofv3"] + ([] if group == "TRACE" else ["--pmc"] + group.replace("@", " ").split()) + ["--kernel-trace", "-d", tmp, "-o", "p", "--", sys.executable,
               os.path.join(REPO, "tools", "ab_tunables.py"), "--cases", a.cases, "--variants", a.variants, "--rounds", str(max(1, a.placements)),
               "--reps", str(a.reps), "--mark", "--shape", a.shape, "--dtype", a.dtype] + (["--realloc"] if a.placements > 1 else [])
        env = dict(os.environ, TMPDIR="/tmp")
        try:  # a counter group the profiler cannot serve must not cost the session (TA_* counters hung a pass for 15 minutes)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=ap_timeout)
            out, rc = r.stdout, r.returncode
        except subprocess.TimeoutExpired as exc:
            out, rc = f"timed out after {ap_timeout} s: {exc.stdout[-300:] if exc.stdout else ''}", -1
        alg = {}  # case -> algorithmic bytes per launch, from the lines ab_tunables prints
        for ln in str(out).splitlines():
            if ln.startswith("{") and '"alg_bytes"' in ln:
                try:
                    d = json.loads(ln)
                    alg[d["case"]] = d["alg_bytes"]
                except ValueError:
                    pass
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if rc != 0 or not dbs:
            print(json.dumps({"pmc": group, "error": str(out)[-600:]}), flush=True)
            shutil.rmtree(tmp, ignore_errors=True)
            continue
        con = sqlite3.connect(dbs[0])
        disp = list(con.execute("select dispatch_id, name, start, duration, grid_x, vgpr_count from kernels order by start"))
        ctr = {}
        for did, cname, val in con.execute("select dispatch_id, counter_name, value from counters_collection"):
            ctr.setdefault(did, {})[cname] = val
        block, first = None, True
        for did, name, _start, dur, gx, vgpr in disp:
            k = short(name)
            if k.startswith("k_fill_synthetic") and gx % 4096 == 0 and gx < 2097152:
                idx = gx // 4096 - 1
                block, first = (idx // len(cases), idx % len(cases)), True
                continue
            if block is None or dur < a.min_us * 1e3:
                continue
            if first:  # cold behind the marker
                first = False
                continue
            e = table.setdefault(block + (k,), {"dur": [], "vgpr": vgpr})
            e["dur"].append(dur / 1e3)
            for cname, val in ctr.get(did, {}).items():
                e.setdefault(cname, []).append(val)
        shutil.rmtree(tmp, ignore_errors=True)
        for (vi, ci, k), e in sorted(table.items()):
            if vi >= len(variants) or ci >= len(cases):
                continue
            d = sorted(e["dur"])
            row = {"pass": group, "variant": variants[vi], "case": cases[ci], "kernel": k[:70], "vgpr": e["vgpr"], "n": len(d),
                   ("us" if group == "TRACE" else "us_under_pmc"): round(d[len(d) // 2], 1) if d else None}
            if group == "TRACE" and d:
                row["us_min"], row["us_max"] = round(d[0], 1), round(d[-1], 1)
            if cases[ci] in alg:
                row["alg_bytes"] = alg[cases[ci]]
            for cname, vals in e.items():
                if cname in ("dur", "vgpr"):
                    continue
                m = sum(vals) / len(vals)
                row[cname] = round(m, 1)
                if cname == "FETCH_SIZE":
                    row["read_GB"] = round(m * 1024 * 2 / 1e9, 3)
                if cname == "WRITE_SIZE":
                    row["write_GB"] = round(m * 1024 / 1e9, 3)
            # occupancy-type counters: <X>_LEVEL accumulates the requests in flight per cycle, so LEVEL / count = mean latency in cycles
            for lvl, cnt, name in (("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "ea_read_latency_cyc"),
                                   ("TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum", "ea_write_latency_cyc"),
                                   ("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "tcp_tcc_read_latency_cyc"),
                                   ("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM", "vmem_inst_latency_cyc")):
                if lvl in row and row.get(cnt):
                    row[name] = round(row[lvl] / row[cnt], 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
