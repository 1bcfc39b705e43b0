#!/bin/bash
# Round 3, session AF: K2Sm (metric stencils along Y, y-stacked workgroups): parity per shape + A/B
S=$PWD/gpurun_out/r03af
mkdir -p $S
export TMPDIR=/tmp
for v in 12 14 18 22 24; do
  echo "== parity met_ys=$v"; XG_MET_YS1=$v XG_MET_YS2=$v timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
done
timeout 400 python tools/ab_tunables.py --cases dY --variants "met_ys1=0;met_ys1=12;met_ys1=14;met_ys1=18;met_ys1=22;met_ys1=24" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_dy.jsonl | cut -c1-150
timeout 400 python tools/ab_tunables.py --cases iYmw --variants "met_ys2=0;met_ys2=12;met_ys2=14;met_ys2=18;met_ys2=22;met_ys2=24" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_iymw.jsonl | cut -c1-150
