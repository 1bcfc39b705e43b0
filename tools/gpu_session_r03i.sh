#!/bin/bash
# Round 3, session I: (a) band heights after the change; (b) what a launch of the Z scan costs beyond its bytes:
# 1 / 2 / 4 / 8 records in one launch, march and level-major
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03i
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases cumZ,cumZr2,cumZr4,cumZr8 --variants "scan_levels=0;scan_levels=1" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_cumZ_records.jsonl | cut -c1-170
timeout 300 python tools/ab_tunables.py --cases i2mw,dY,iYmw,iXmw --variants "zb_rows=16;zb_rows=8;zb_rows=32" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_bands.jsonl | cut -c1-170
timeout 300 python tools/pmc_ab.py --cases i2mw,dY,iYmw,iXmw --variants "zb_rows=16;zb_rows=8" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_bands.jsonl | cut -c1-250
