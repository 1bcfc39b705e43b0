#!/bin/bash
# Round 3, session AG: K2Sm with two metrics: larger tasks
S=$PWD/gpurun_out/r03ag
mkdir -p $S
export TMPDIR=/tmp
for v in 28 42 44; do
  echo "== parity met_ys=$v"; XG_MET_YS1=$v XG_MET_YS2=$v timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -1 | tee -a $S/pytest.log
done
timeout 400 python tools/ab_tunables.py --cases iYmw --variants "met_ys2=0;met_ys2=18;met_ys2=28;met_ys2=42;met_ys2=44" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_iymw.jsonl | cut -c1-150
timeout 400 python tools/ab_tunables.py --cases dY,dZ,iZmw --variants "met_ys1=0;met_ys1=12" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_dy.jsonl | cut -c1-150
