#!/usr/bin/env python3
"""TCC counters of the store-pattern probe's variants (tools/probes/store_probe.hip, section `pmc`): one `rocprofv3 --pmc` pass with
--kernel-trace only, dispatches joined to the probe's own lines by ORDER (every line = 9 dispatches of one kernel; the k_diffcount /
memset helpers are skipped by name).

    python tools/probes/store_probe_pmc.py [--counters TCC_TAG_STALL,TCC_EA0_RDREQ_DRAM_CREDIT_STALL,TCC_EA0_WRREQ_STALL,TCC_BUSY]
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--counters", default="TCC_TAG_STALL,TCC_EA0_RDREQ_DRAM_CREDIT_STALL,TCC_EA0_WRREQ_STALL,TCC_BUSY")
    a = ap.parse_args()
    counters = a.counters.split(",")
    tmp = tempfile.mkdtemp(prefix="xg_sp_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", tmp, "-o", "p", "--", os.path.join(REPO, "tools", "probes", "store_probe"), "75", "2400", "3600", "pmc"]
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=280)
        lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{") and "variant" in ln]
        dbs = sorted(glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True))
        if p.returncode != 0 or not dbs:
            print(json.dumps({"error": f"rocprofv3 rc {p.returncode}", "stderr": p.stderr[-400:]}))
            return
        db = sqlite3.connect(dbs[0])
        rows = db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id").fetchall()
        per = {}
        for did, name, cname, val in rows:
            if not any(k in name for k in ("k_flat", "k_pattern", "k_levelchain")):
                continue
            per.setdefault(did, {"kernel": name.split("(")[0][:60]})
            per[did][cname] = per[did].get(cname, 0.0) + val
        disp = [per[k] for k in sorted(per)]
        if len(disp) != 9 * len(lines):
            print(json.dumps({"error": f"{len(disp)} dispatches for {len(lines)} probe lines"}))
        for i, ln in enumerate(lines):
            grp = disp[9 * i + 6: 9 * i + 9]
            if not grp:
                break
            ln["kernel"] = grp[0]["kernel"]
            ln["counters_per_launch"] = {c: round(sum(g.get(c, 0.0) for g in grp) / len(grp), 1) for c in counters}
            ln["ms_note"] = "ms measured under the counter pass (serialised dispatches)"
            print(json.dumps(ln), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
