#!/bin/bash
# Round-2 tunable sweeps on one GPU (each point = one process: the library reads its tunables once).
# Usage on the GPU box: bash tools/sweep_r02.sh <part> > gpurun_out/sweep_<part>.jsonl
PART=${1:-met}
MB="python tools/microbench.py --reps 15"
run() { env "$@" $MB --cases "$CASES" 2>/dev/null | grep '^{'; }
if [ "$PART" = met ]; then
  CASES=metric
  run XG_CONTIG_RW=0 XG_MET_SEG=1
  run XG_CONTIG_RW=1 XG_MET_SEG=1
  run XG_CONTIG_RW=2 XG_MET_SEG=2
  run XG_CONTIG_RW=4 XG_MET_SEG=4
  run XG_CONTIG_RW=2 XG_MET_SEG=2 XG_ZB_ROWS=32
  run XG_CONTIG_RW=2 XG_MET_SEG=2 XG_ZB_ROWS=8
  run XG_CONTIG_RW=2 XG_MET_SEG=2 XG_ZBAND=0
  run XG_CONTIG_RW=0 XG_MET_SEG=1
  run XG_CONTIG_RW=2 XG_MET_SEG=2
fi
if [ "$PART" = scan ]; then
  CASES=scanY
  run XG_SCAN_PIPE=0
  for u in 8 16 24 32; do run XG_SCAN_PIPE=1 XG_SCAN_U=$u; done
  for u in 8 16 24 32; do run XG_SCAN_PIPE=1 XG_SCAN_U=$u XG_SCAN_NARROW_BELOW=0; done
  for u in 16 32; do run XG_SCAN_PIPE=1 XG_SCAN_U=$u XG_SCAN_PACE=1; done
  run XG_SCAN_PIPE=1 XG_SCAN_U=16 XG_SCAN_PACE=1 XG_SCAN_NARROW_BELOW=0
  run XG_SCAN_PIPE=0 XG_SCAN_NARROW_BELOW=0
  run XG_SCAN_PIPE=1 XG_SCAN_U=16 XG_MARCH_BAND=0
  CASES=scanZ
  run XG_SCAN_PIPE=0
  run XG_SCAN_PIPE=1
  run XG_SCAN_PIPE=2
  run XG_SCAN_PIPE=0
fi
