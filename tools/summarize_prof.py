#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


for f in find(os.path.join(out, f"prof_stats_{tag}"), "*kernel_stats.csv"):
    print(f"# {os.path.relpath(f, out)}")
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print(",".join(c[:70] for c in row))

for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in find(os.path.join(out, f"prof_{which}_{tag}"), "*counter_collection.csv"):
        agg = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                if row.get("Counter_Name") != counter:
                    continue
                k = row.get("Kernel_Name", "?")[:90]
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0))
        print(f"# {counter} per dispatch (raw counter units, mean) from {os.path.relpath(f, out)}")
        for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"{counter},{n},{tot / n:.6g},{k}")
