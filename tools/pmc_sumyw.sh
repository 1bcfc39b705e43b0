#!/bin/bash
# where do the weights of a weighted march come from?  FETCH_SIZE (HBM reads) and L2 hits / misses of sum(T * w) along Y:
# the march (scan_chain=0) against the chained chunks K4c (scan_chain=1), dy(Y, X) shared by the 75 levels
cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in "scan_chain=0" "scan_chain=1" "scan_chain=1,scan_chain_w=101"; do
  for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pmcx
    timeout 100 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmcx -o p -- python $REPO/tools/ab_tunables.py --cases sumYw --rounds 1 --reps 2 --variants "$v" > /tmp/pmcx.log 2>&1 || { echo "rocprofv3 failed / timed out for $ctr"; tail -3 /tmp/pmcx.log; continue; }
    python - "$v" <<'PY'
import sqlite3, sys, glob
dbs = glob.glob("/tmp/pmcx/**/*.db", recursive=True)
if not dbs:
    print("no db"); sys.exit(0)
con = sqlite3.connect(dbs[0])
dur = dict(con.execute("select dispatch_id, duration from kernels").fetchall())
rows = con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection where kernel_name like '%reduce_%'").fetchall()
agg = {}
for did, k, c, v in rows:
    agg.setdefault((k.split('(')[0][-44:], did), {})[c] = v
seen = set()
for (k, did), d in sorted(agg.items()):
    key = (k, tuple(round(x, -5) for x in d.values()))
    if key in seen: continue
    seen.add(key)
    print(sys.argv[1], k, f"{dur.get(did,0)/1e6:.3f} ms", {c: round(v/1e6, 2) for c, v in d.items()})
PY
  done
done
