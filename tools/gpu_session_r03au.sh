#!/bin/bash
# Round 3, session AU: K2Sy with two x-tiles per wave (the scalar row logic paid once per 2 KB)
S=$PWD/gpurun_out/r03au
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; XG_SEG_YS=2 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases diffY,diffX --variants "seg_ys=0;seg_ys=1;seg_ys=2" --rounds 8 --reps 7 2>&1 | grep '^{' | tee $S/ab_k2sy_xt.jsonl | cut -c1-160
