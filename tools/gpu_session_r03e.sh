#!/bin/bash
# Round 3, session E: what do the slow kernels wait for?  Latency / stall / TLB counters of a flat stencil (diffX), a
# metric-carrying one (dY), the Z scan as a march and level-major, and the chained Y scan -- one table per box; together
# with box_probe's rates this is also the "two kinds of boxes" table.
# usage: bash tools/gpu_session_r03e.sh <tag> "<counter groups separated by |>"
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/${1:-r03e}
mkdir -p $S
export TMPDIR=/tmp
GROUPS_=${2:-"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum|TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum|SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"}
bash tools/box_probe.sh 2>&1 | grep "rates" | tee $S/box_rates.txt
SECONDS=0
timeout 900 python tools/pmc_ab.py --cases ${3:-diffX,dY,cumZ,cumY} --variants "${4:-scan_levels=0;scan_levels=1}" --reps 2 --pass-timeout 120 --pmc "$GROUPS_" 2>&1 | tee -a $S/pmc_wait.jsonl | cut -c1-100
echo "pmc passes took $SECONDS s"
