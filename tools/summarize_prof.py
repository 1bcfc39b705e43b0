#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) outputs into the text summary committed under profiles/.

    python tools/summarize_prof.py gpurun_out r01 > profiles/history/r01_rocprof_summary.txt

Kernel stats come from the --kernel-trace --stats run; FETCH_SIZE / WRITE_SIZE from two separate
--pmc passes.  Counter units: KiB.  On gfx950 FETCH_SIZE reports exactly half of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section), so fetch bytes = FETCH_SIZE * 1024 * 2;
WRITE_SIZE * 1024 is calibrated here against k_fill_synthetic, which writes exactly n*8 bytes.
"""
import glob
import json
import os
import sqlite3
import sys

out, tag = sys.argv[1], sys.argv[2]


def db(sub):
    hits = sorted(glob.glob(os.path.join(out, f"prof_{sub}_{tag}", "**", "*.db"), recursive=True))
    return sqlite3.connect(hits[0]) if hits else None


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


con = db("stats")
full = {}
if con:
    print("# kernel-trace: per-kernel launches (full-size launches only: duration > 0.5 ms), durations in us")
    print("kernel,calls,avg_us,min_us,max_us")
    q = ("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels "
         "where duration > 500000 group by name order by sum(duration) desc")
    for name, n, avg, mn, mx in con.execute(q):
        print(f"{short(name)},{n},{avg:.1f},{mn:.1f},{mx:.1f}")
        full[short(name)] = avg
    print("# kernel-trace: all kernels (top_kernels view)")
    for name, n, tot, avg, pct in con.execute("select * from top_kernels limit 10"):
        print(f"{short(name)},{n},total_us={tot:.1f},avg_us={avg:.1f},{pct:.2f}%")

traffic = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    con = db(sub)
    if not con:
        continue
    print(f"# pmc pass: {counter} per dispatch in KiB (full-size launches: value > 1e6)")
    q = ("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
         f"where counter_name='{counter}' and value > 1e6 group by kernel_name order by avg(value) desc")
    for name, n, avg, mn, mx in con.execute(q):
        print(f"{short(name)},{n},{avg:.1f},{mn:.1f},{mx:.1f}")
        traffic.setdefault(short(name), {})[counter] = avg

print("# HBM traffic per full-size launch: FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024, bytes")
summary = {}
for k, v in traffic.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
        summary[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
        print(f"{k},read={rd:.4g},write={wr:.4g},total={rd + wr:.5g}")
by_base = {}
for k, v in summary.items():
    by_base.setdefault(k.split("<")[0], []).append(v["total_bytes"])
flat = {"_comment": "HBM bytes per full-size launch of the bench workload (3600x2400x75 f64), mean over the template "
                    "instances of each kernel; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md",
        "_source": f"profiles/{tag.split('_')[0]}_rocprof_summary_{'_'.join(tag.split('_')[1:]) or 'bench'}.txt"}
flat.update({k: sum(v) / len(v) for k, v in by_base.items()})
with open(os.path.join(out, f"pmc_traffic_{tag}.json"), "w") as f:
    json.dump(flat, f, indent=1)
