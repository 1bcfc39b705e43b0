#!/bin/bash
# Round 3, session BJ: `sc1 nt` stores in the general-length stencil kernels (K1g, K2g) and the metric / two-axis kernels that
# kept `nt`: the stencil unit with -DXG_STG_DROP_ALL against the product, alternating processes
S=$PWD/gpurun_out/r03bj
mkdir -p $S
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in product dropall; do
  if [ $lib = product ]; then unset XG_HIP_LIB; else export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_$lib.so; fi
  timeout 300 python tools/ab_tunables.py --cases diffXo,diffYo,iYmw,i2mw --variants "nt_store=1" --rounds 3 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r[d['case']]=d['frac_8TBps']
print(json.dumps({'lib':'$lib','rep':$rep, **r}))
" | tee -a $S/ab_drop_general_kernels.jsonl
done; done
