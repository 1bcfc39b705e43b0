#!/bin/bash
# Round 3, session BL: gradient (two outputs): `sc1 nt` on one of the two output streams, or both, against `nt` on both
S=$PWD/gpurun_out/r03bl
mkdir -p $S
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in product g1 g2 g3; do
  if [ $lib = product ]; then unset XG_HIP_LIB; else export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_$lib.so; fi
  timeout 300 python tools/ab_tunables.py --cases grad --variants "nt_store=1" --rounds 4 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'lib':'$lib','rep':$rep,'grad':d['frac_8TBps'],'min_ms':d['min_ms'],'max_ms':d['max_ms']}))
" | tee -a $S/ab_grad_drop.jsonl
done; done
