#!/bin/bash
# Round 3, session BC: `sc1 nt` stores in the flat X kernel (no metrics) and K2Sy only: parity, then alternating processes against
# the nt-only library (bench.py and the two plain stencils)
S=$PWD/gpurun_out/r03bc
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee $S/pytest.log
for rep in 1 2 3 4; do
for lib in ntonly new; do
  if [ $lib = new ]; then unset XG_HIP_LIB; else export XG_HIP_LIB=$PWD/tools/_ab_libs/libxgcm_hip_$lib.so; fi
  timeout 300 python tools/ab_tunables.py --cases diffX,diffY,dY --variants "nt_store=1" --rounds 3 --reps 5 2>&1 | grep '^{' | python -c "
import sys,json
r={}
for l in sys.stdin:
    d=json.loads(l); r[d['case']]=d['frac_8TBps']
print(json.dumps({'lib':'$lib','rep':$rep, **r}))
" | tee -a $S/ab_store_policy_plain.jsonl
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','rep':$rep,'bench':d['value'],'frac':d['roofline']['frac'],'per_op_ms':d['roofline']['per_op_ms']}))" | tee -a $S/bench_store_policy.jsonl
done; done
