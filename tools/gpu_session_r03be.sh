#!/bin/bash
# Round 3, session BE: (1) parity of the build with `sc1 nt` in K1r / K2S / K2Sm / the two-axis kernel without metrics; (2) the
# vector unit with `sc1 nt` everywhere (experiment library) against the product, alternating processes
S=$PWD/gpurun_out/r03be
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee $S/pytest.log
for rep in 1 2 3; do
for lib in product dropall; do
  if [ $lib = product ]; then unset XG_HIP_LIB; else export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_$lib.so; fi
  timeout 300 python tools/ab_tunables.py --cases divg,vort,divT,mulTT,flux,grad --variants "nt_store=1" --rounds 3 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r[d['case']]=d['frac_8TBps']
print(json.dumps({'lib':'$lib','rep':$rep, **r}))
" | tee -a $S/ab_drop_vector_unit.jsonl
done; done
