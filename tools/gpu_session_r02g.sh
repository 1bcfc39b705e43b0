#!/bin/bash
# Round 2, session G: what bounds the linear transform on random columns and the Y scan -- PMC counters
OUT=$PWD/gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -c . $OUT/counters_avail.txt
CMD="python $REPO/tools/ab_tunables.py --cases tlin_rw,tlin_sm,tcon_rw,cumY,cumZ,diffY --rounds 1 --reps 2 --variants dbg=0"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA_WR_UNCACHED_32B_sum TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_DRAM_sum TCC_EA_WRREQ_DRAM_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_$tag -o pmc -- $CMD > $OUT/pmc_$tag.log 2>&1
  db=$(find $OUT/pmc_$tag -name "*.db" | head -1)
  echo "== $set"
  [ -n "$db" ] && python - "$db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
dur = dict(con.execute("select name, avg(duration) from kernels group by name").fetchall())
for k, c, v, n in rows:
    short = k.replace("void (anonymous namespace)::", "").split("(")[0]
    if dur.get(k, 0) > 5e5:
        print(f"{short[:60]:60s} {c:28s} {v:16.1f}  n={n} avg_ms={dur[k]/1e6:.3f}")
PY
  tail -2 $OUT/pmc_$tag.log | head -1
done
