#!/bin/bash
# Round 3, session BA: cache policy of the streaming output stores: nt (the product) against sc1, sc0 sc1 and nt sc1 (libraries
# built with -DXG_STORE_POLICY=1/2/3 for the f64 stencil / scan / vector units), alternating processes, three rounds
S=$PWD/gpurun_out/r03ba
mkdir -p $S
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in nt pol1 pol2 pol3; do
  if [ $lib = nt ]; then unset XG_HIP_LIB; else export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_$lib.so; fi
  timeout 300 python tools/ab_tunables.py --cases diffX,diffY,dY,cumZ,mulTT,grad --variants "nt_store=1" --rounds 3 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r[d['case']]=d['frac_8TBps']
print(json.dumps({'lib':'$lib','rep':$rep, **r}))
" | tee -a $S/ab_store_policy.jsonl
done; done
