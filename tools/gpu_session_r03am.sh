#!/bin/bash
# Round 3, session AM: K4L with level groups fastest (an XCD band = all level groups of a few x-tiles)
S=$PWD/gpurun_out/r03am
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; XG_REDUCE_LDSW=2 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases sumYw,avgYw --variants "reduce_ldsw=1;reduce_ldsw=2" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_k4l_order.jsonl | cut -c1-150
timeout 300 python tools/pmc_ab.py --cases sumYw --variants "reduce_ldsw=1;reduce_ldsw=2" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_k4l_order.jsonl | cut -c1-300
