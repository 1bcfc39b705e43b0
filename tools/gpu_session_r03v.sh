#!/bin/bash
# Round 3, session V: chained scan with level-quad workgroups (metric rows through LDS): parity + A/B
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03v
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_chain_rescue.py tests/test_gpu_graphs.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases cumYw,cumY --variants "scan_chain_lq=0;scan_chain_lq=1" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_lq.jsonl | cut -c1-150
timeout 300 python tools/pmc_ab.py --cases cumYw --variants "scan_chain_lq=0;scan_chain_lq=1" --pmc "FETCH_SIZE|WRITE_SIZE" 2>&1 | tee $S/pmc_lq.jsonl | cut -c1-250
