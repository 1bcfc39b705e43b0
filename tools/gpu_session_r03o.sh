#!/bin/bash
# Round 3, session O: chained kernels with a metric shared by the levels, columns numbered x-tile-major
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03o
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_chain_rescue.py tests/test_gpu_graphs.py -x -q 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases cumYw,sumYw,cumY --variants "scan_chain_tmaj=0;scan_chain_tmaj=1" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_tmaj.jsonl | cut -c1-150
timeout 300 python tools/pmc_ab.py --cases cumYw,sumYw --variants "scan_chain_tmaj=0;scan_chain_tmaj=1" --pmc "FETCH_SIZE|WRITE_SIZE" 2>&1 | tee $S/pmc_tmaj.jsonl | cut -c1-260
