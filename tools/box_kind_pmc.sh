#!/bin/bash
# One sample for the "two kinds of boxes" table: this box's steady rates next to the counters that
# could tell the kinds apart -- request latency at the L2's fabric interface (TCC_EA0_*_LEVEL / requests), fabric credit
# stalls, the L1 TLB, wave-level wait cycles.  usage: bash tools/box_kind_pmc.sh <out-file>
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUTF=${1:-$REPO/gpurun_out/box_kind.txt}
export TMPDIR=/tmp
{
  echo "# $(date -u +%FT%TZ) $(rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1 | tr -s ' ')  $(rocm-smi --showbus 2>/dev/null | grep -i "PCI Bus" | head -1 | tr -s ' ')"
  bash $REPO/tools/box_probe.sh 2>&1 | grep "rates"
  timeout 300 python $REPO/tools/pmc_ab.py --cases diffX,dY,cumZ,cumY --variants "scan_chain=1" --reps 2 --pass-timeout 100 --pmc "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum|TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum|TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum|SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" 2>&1
} | tee -a $OUTF | cut -c1-160
