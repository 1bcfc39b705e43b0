#!/bin/bash
# Round 3, session AI: K2Sm as the default for one metric: which kernel runs, its traffic, A/B against K2S
S=$PWD/gpurun_out/r03ai
mkdir -p $S
export TMPDIR=/tmp
timeout 400 python tools/ab_tunables.py --cases dY --variants "met_ys1=0;met_ys1=12;met_ys1=14" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_dy.jsonl | cut -c1-150
timeout 300 python tools/pmc_ab.py --cases dY --variants "met_ys1=12;met_ys1=0" --pmc "FETCH_SIZE|TCP_TCC_READ_REQ_sum" 2>&1 | tee $S/pmc_dy.jsonl | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "y_stacked" 2>&1 | tail -2 | tee $S/pytest.log
