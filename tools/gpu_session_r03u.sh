#!/bin/bash
# Round 3, session U: K4L (weighted reduction, weights through LDS): parity + A/B against the chained kernel
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03u
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_chain_rescue.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases sumYw,avgYw,sumY --variants "reduce_ldsw=0;reduce_ldsw=1" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_ldsw.jsonl | cut -c1-150
timeout 300 python tools/pmc_ab.py --cases sumYw,avgYw --variants "reduce_ldsw=1" --pmc "FETCH_SIZE|WRITE_SIZE" 2>&1 | tee $S/pmc_ldsw.jsonl | cut -c1-250
