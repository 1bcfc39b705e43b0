#!/usr/bin/env python3
"""VALU issue pressure per kernel from a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace run (rocpd .db).

    python tools/valu_util.py gpurun_out/prof_valu/**/x_results.db

Prints, per kernel (full-size launches only): wave-instructions per wave and the share of the kernel's
duration that 1024 SIMD16 units need to issue them (4 cycles per wave64 VALU instruction at `--mhz`).
A kernel whose share approaches 1 is bound by its own instruction stream, not by HBM."""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--mhz", type=float, default=2400.0)
ap.add_argument("--simds", type=int, default=1024)
a = ap.parse_args()
con = sqlite3.connect(a.db)
q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection where dispatch_id in "
     "(select dispatch_id from counters_collection where counter_name='SQ_WAVES' and value > 20000) "
     "group by kernel_name, counter_name")
vals = {}
for name, ctr, v, n in con.execute(q):
    vals.setdefault(name, {})[ctr] = v
dur = {name: (avg, n) for name, avg, n in con.execute(
    "select name, avg(duration), count(*) from kernels where duration > 300000 group by name")}
print("kernel,avg_ms,valu_per_wave,salu_per_wave,valu_issue_share")
for name, c in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if name not in dur or "SQ_INSTS_VALU" not in c:
        continue
    ms = dur[name][0] / 1e6
    waves = c.get("SQ_WAVES", 0) or 1
    valu = c["SQ_INSTS_VALU"]
    share = valu * 4 / a.simds / (ms * 1e-3 * a.mhz * 1e6)
    short = name.replace("void (anonymous namespace)::", "").split("(")[0]
    print(f"{short},{ms:.3f},{valu / waves:.0f},{c.get('SQ_INSTS_SALU', 0) / waves:.0f},{share:.2f}")
