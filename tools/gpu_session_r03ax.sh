#!/bin/bash
# Round 3, session AX: 32-bit index arithmetic in K2Sy / K2Sm: the library before the change (tools/_ab_libs/libxgcm_hip_old.so,
# XG_HIP_LIB) and after it, alternating; within each process the y-stacked kernels are timed against K2S (unchanged), so the
# in-process difference to K2S is what compares across the two libraries
S=$PWD/gpurun_out/r03ax
mkdir -p $S
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in old new; do
  if [ $lib = old ]; then export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_old.so; else unset XG_HIP_LIB; fi
  timeout 300 python tools/ab_tunables.py --cases diffY,dY --variants "seg_ys=0,met_ys1=0;seg_ys=1,met_ys1=12" --rounds 6 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r[(d['case'],d['variant'])]=d['frac_8TBps']
for c in ('diffY','dY'):
    a=r[(c,'seg_ys=0,met_ys1=0')]; b=r[(c,'seg_ys=1,met_ys1=12')]
    print(json.dumps({'lib':'$lib','case':c,'K2S':a,'ystacked':b,'delta':round(b-a,4)}))
" | tee -a $S/ab_lean32_old_new.jsonl
done; done
