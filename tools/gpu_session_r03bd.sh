#!/bin/bash
# Round 3, session BD: `sc1 nt` for EVERY 16-B non-temporal store of the stencil, scan and pad units (experiment library,
# -DXG_STG_DROP_ALL) against the product, alternating processes: which other single-output kernels gain?
S=$PWD/gpurun_out/r03bd
mkdir -p $S
export TMPDIR=/tmp
for rep in 1 2 3 4; do
for lib in product dropall; do
  if [ $lib = product ]; then unset XG_HIP_LIB; else export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_$lib.so; fi
  timeout 300 python tools/ab_tunables.py --cases dY,iXmw,iYmw,i2,i2mw,dX --variants "nt_store=1" --rounds 3 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r[d['case']]=d['frac_8TBps']
print(json.dumps({'lib':'$lib','rep':$rep, **r}))
" | tee -a $S/ab_drop_stencil_unit.jsonl
done; done
